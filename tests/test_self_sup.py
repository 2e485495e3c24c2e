"""NSA / CutPaste self-supervised augmentation (reference minigpt4/datasets/self_sup_tasks.py:11-292).
CPU: the oracle restatement against the golden produced by the reference's own `patch_ex`; the product's host path against
the oracle.  GPU: the blend / label kernels against the oracle."""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import self_sup_ref as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "self_sup.npz"))
_spec = importlib.util.spec_from_file_location("mk_selfsup", os.path.join(os.path.dirname(os.path.dirname(__file__)), "tools",
                                                                         "make_golden_selfsup.py"))
MK = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MK)            # CASES + the synthetic test images (pure numpy; the reference is NOT imported)


@pytest.mark.parametrize("name,seed,kw", MK.CASES, ids=[c[0] for c in MK.CASES])
def test_oracle_matches_the_reference_patch_ex(name, seed, kw):
    dest, src = MK.test_image(10 + seed), MK.test_image(40 + seed)
    np.random.seed(1000 + seed)
    patchex, label, boxes = O.patch_ex(dest.copy(), src.copy(), **kw)
    assert np.random.randint(1 << 30) == int(G[name + "_next_draw"])          # same number of random draws
    assert np.array_equal(patchex, G[name + "_patchex"])                      # bit-exact uint8 image
    assert np.array_equal(np.asarray(boxes, np.int64).reshape(-1, 4), G[name + "_boxes"])
    np.testing.assert_allclose(np.asarray(label, np.float64), G[name + "_label"], rtol=0, atol=1e-12)


from myriad_amd import self_sup as P  # noqa: E402


@pytest.mark.parametrize("name,seed,kw", MK.CASES, ids=[c[0] for c in MK.CASES])
def test_product_host_path_matches_the_oracle(name, seed, kw):
    """myriad_amd.self_sup.plan + apply_numpy (what a DataLoader worker runs) == the oracle == the reference's patch_ex."""
    dest, src = MK.test_image(10 + seed), MK.test_image(40 + seed)
    np.random.seed(1000 + seed)
    patchex, label, boxes = P.patch_ex(dest.copy(), src.copy(), **kw)
    assert np.random.randint(1 << 30) == int(G[name + "_next_draw"])
    assert np.array_equal(patchex, G[name + "_patchex"])
    assert np.array_equal(np.asarray(boxes, np.int64).reshape(-1, 4), G[name + "_boxes"])
    np.testing.assert_allclose(np.asarray(label, np.float64), G[name + "_label"], rtol=0, atol=1e-12)
    with pytest.raises(NotImplementedError):
        P.patch_ex(dest, src, mode="poisson")


@pytest.mark.gpu
def test_device_blend_and_label_match_the_oracle():
    """csrc/selfsup.hip on a batch of crops in HBM: bit-exact uint8 images, labels within fp32 rounding of the float64 oracle,
    for every blend / label mode of the golden cases (one batch per label mode: the label kernel takes one mode per launch)."""
    ex = P.PatchExHIP("cuda")
    by_mode = {}
    for name, seed, kw in MK.CASES:
        by_mode.setdefault((kw.get("label_mode", "binary"), kw.get("intensity_logistic_params", (1 / 6, 20))), []).append((name, seed, kw))
    for (label_mode, ilp), cases in by_mode.items():
        dests, srcs, plans = [], [], []
        for name, seed, kw in cases:
            dest, src = MK.test_image(10 + seed), MK.test_image(40 + seed)
            np.random.seed(1000 + seed)
            pk = {k: v for k, v in kw.items() if k != "intensity_logistic_params"}
            ops, factor = P.plan(dest, src, **pk)
            dests.append(dest)
            srcs.append(dest if kw.get("same") else src)
            plans.append((ops, factor))
        d = torch.from_numpy(np.stack(dests)).cuda()
        s_ = torch.from_numpy(np.stack(srcs)).cuda()
        out, label, union = ex(d, s_, plans, label_mode=label_mode, intensity_logistic_params=ilp)
        for i, (name, seed, kw) in enumerate(cases):
            assert np.array_equal(out[i].cpu().numpy(), G[name + "_patchex"]), name
            np.testing.assert_allclose(label[i].cpu().numpy().astype(np.float64), G[name + "_label"][..., 0], rtol=0, atol=2e-5, err_msg=name)
