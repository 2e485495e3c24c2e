"""SURVEY 8 f-1 on the GPU: the HIP vision expert against the goldens the reference's own code produced and against the
fp32 oracle on the same seeded weights.  bf16 GEMM operands, fp32 accumulation: tolerances stated per quantity."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from myriad_amd import ops  # noqa: E402
from myriad_amd.vision_expert import VisionExpertHIP  # noqa: E402
from oracle import expert_ref as X  # noqa: E402
from tests import golden_utils as gu  # noqa: E402
from tests.test_expert_oracle import load_case  # noqa: E402

DEV = "cuda"


def relerr(a, b):
    a, b = a.double().cpu(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.mark.parametrize("name", ["d1280_3blk", "d1280_1blk_k2"])
def test_vision_expert_vs_reference_golden(name):
    g, cfg, sd, images, refs, text = load_case(name)
    ex = VisionExpertHIP(sd, cfg["heads"], cfg["layers"], DEV)
    emb, taps = ex.trunk.forward(images.to(DEV), want_embedding=True)
    for i, t in enumerate(taps):
        assert relerr(t[:, ::8, ::16], g[f"tap{i}_sub"]) < 2e-2, i            # activations: 2e-2 of max-abs
    assert (emb.cpu() - torch.from_numpy(g["image_embeds"])).abs().max() < 2e-2   # unit vectors
    zmap, zmask = ex.zero_shot(images, text)
    # probabilities in [0,1]: logits are 100 * cosine, so a 2e-3 cosine error moves a mid-range probability by ~5e-2
    assert (zmap.cpu() - torch.from_numpy(g["zs_map"])).abs().max() < 8e-2
    assert (zmask.cpu() - torch.from_numpy(g["zs_mask"])).abs().max() < 8e-2
    assert (zmap.cpu() - torch.from_numpy(g["zs_map"])).abs().mean() < 1e-2
    omap, omask = ex.one_shot(images, refs)
    assert (omap.cpu() - torch.from_numpy(g["os_map"])).abs().max() < 1e-2       # 1 - cosine similarity
    assert (omask.cpu() - torch.from_numpy(g["os_mask"])).abs().max() < 1e-2
    assert zmap.shape == (cfg["B"], 1, 224, 224) and omask.shape == (cfg["B"], 1, 16, 16)
    # one trunk pass over [images ; references] must give the same four tensors (batch rows are independent)
    (zmap2, zmask2), (omap2, omask2) = ex.forward(images, text, refs)
    # ... up to the GEMM policy: with a split-K scratch registered the K split count depends on M, the fp32 partial sums are
    # added in a different order, and 100 * cosine logits amplify a flipped bf16 rounding just as they do against the golden
    # above (same tolerances; bit-equal when no scratch is registered, i.e. when this file runs alone)
    for a, b in ((zmap, zmap2), (zmask, zmask2)):
        assert (a - b).abs().max() < 8e-2 and (a - b).abs().mean() < 5e-3
    for a, b in ((omap, omap2), (omask, omask2)):
        assert (a - b).abs().max() < 1e-2


def test_vision_expert_full_depth_vs_oracle():
    """The full ImageBind-Huge vision trunk (32 blocks x 1280, taps after blocks 8/16/24/32 as adrefexpert_v2.py:16-29 taps
    them) and both map heads against the oracle (pinned to the reference classes at reduced depth by the goldens above) on
    the same seeded weights: the depth the goldens cannot afford."""
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    D, heads, blocks, C, B, k, seed = 1280, 16, 32, 1024, 2, 1, 4242
    layers = [7, 15, 23, 31]
    sd = gu.expert_weights(D, blocks, C, len(layers), seed)
    images, refs, text = gu.expert_inputs(B, k, C, seed + 100)
    emb_o, taps_o = X.vision_trunk(sd, images, heads, layers, blocks)
    _, rtaps_o = X.vision_trunk(sd, refs, heads, layers, blocks)
    dec = {kk[len("image_decoder."):]: v for kk, v in sd.items() if kk.startswith("image_decoder.")}
    zmap_o, zmask_o = X.zero_shot_maps(taps_o, dec, text)
    omap_o, omask_o = X.one_shot_maps(taps_o, rtaps_o)
    ex = VisionExpertHIP(sd, heads, layers, DEV)
    emb, taps = ex.trunk.forward(images.to(DEV), want_embedding=True)
    errs = [relerr(t, to) for t, to in zip(taps, taps_o)]
    print("tap errors", errs, "embedding", (emb.cpu() - emb_o).abs().max().item())
    for i, e in enumerate(errs):
        assert e < 2e-2, (i, e)                                   # activations: 2e-2 of max-abs (measured 5e-3 at full depth)
    assert (emb.cpu() - emb_o).abs().max() < 2e-2                # unit vectors
    # the zero-shot head compared where nothing amplifies the error: the pair logits 100 * cos(patch, text) per tap, in units
    # of the cosine (|d cos| <= 3e-3 at full depth; the maps below push the same error through a softmax of slope 25)
    kept = []
    ex._zero_shot_from_taps(taps, text, keep_logits=kept)
    for lg, lo in zip(kept, X.zero_shot_logits(taps_o, dec, text)):
        assert lg.shape[1] == 2 and lg.shape[0] == lo.shape[0] * lo.shape[1]
        assert float((lg.cpu().view_as(lo) - lo).abs().max()) / 100.0 < 3e-3
    (zmap, zmask), (omap, omask) = ex.forward(images, text, refs)
    assert (zmap.cpu() - zmap_o).abs().max() < 1e-1 and (zmap.cpu() - zmap_o).abs().mean() < 1e-2
    assert (zmask.cpu() - zmask_o).abs().max() < 1e-1
    assert (omap.cpu() - omap_o).abs().max() < 2e-2 and (omask.cpu() - omask_o).abs().max() < 2e-2


def test_map_head_kernels_vs_torch():
    """The fp32 head kernels on their own against torch (tight tolerances: no bf16 involved)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 128, generator=g).to(DEV)
    yb, yf = ops.l2norm_rows(x, want_f32=True)
    ref = x / x.norm(dim=-1, keepdim=True)
    assert relerr(yf, ref.cpu()) < 1e-6 and relerr(yb.float(), ref.cpu()) < 5e-3
    B, L, C = 3, 256, 64
    p = torch.randn(B * L, C, generator=g).to(DEV)
    text = torch.randn(B, 2, C, generator=g)
    text = (text / text.norm(dim=-1, keepdim=True)).to(DEV)
    lg = ops.pair_logits(p, text, L, 100.0)
    pn = (p / p.norm(dim=-1, keepdim=True)).view(B, L, C)
    ref = 100.0 * pn @ text.transpose(-1, -2)
    assert relerr(lg.view(B, L, 2), ref.cpu()) < 1e-5
    mask = torch.zeros(B, 16, 16, device=DEV)
    amap = torch.zeros(B, 224, 224, device=DEV)
    ops.zs_accumulate(lg, mask, amap, 0.5)
    ops.zs_accumulate(lg, mask, amap, 0.5)
    grid = ref.permute(0, 2, 1).reshape(B, 2, 16, 16)
    assert relerr(mask, torch.softmax(grid, 1)[:, 1].cpu()) < 1e-5
    up = torch.nn.functional.interpolate(grid, size=224, mode="bilinear", align_corners=True)
    assert relerr(amap, torch.softmax(up, 1)[:, 1].cpu()) < 1e-4
    s = torch.randn(50, 514, generator=g).to(DEV)
    acc = torch.ones(50, device=DEV)
    ops.rowmax_skip(s, acc, 257, 0.25)
    keep = torch.ones(514, dtype=torch.bool)
    keep[0] = keep[257] = False
    assert relerr(acc, (1 + 0.25 * s[:, keep.to(DEV)].max(dim=1).values).cpu()) < 1e-6
    sim = torch.rand(2, 16, 16, generator=g).to(DEV)
    up = ops.bilinear_ac(sim, 224, 224, one_minus=True)
    ref = 1 - torch.nn.functional.interpolate(sim[:, None], size=224, mode="bilinear", align_corners=True)[:, 0]
    assert relerr(up, ref.cpu()) < 1e-5
    assert relerr(ops.bilinear_ac(sim, 16, 16), sim.cpu()) < 1e-7


def test_model_with_attached_expert_equals_precomputed_maps():
    """`attach_vision_expert`: maps produced inside the model from (image, text pair, references) give exactly the loss
    of the same maps handed in through samples (myriad.py:331-345 vs the maps-as-inputs boundary)."""
    from myriad_amd.myriad import MyriadHIP
    from tests.test_model_gpu import _composite_sd
    sd = _composite_sd([1, 2, 3, 4, 5])
    image, _, before, after, tgt, tmask = gu.synthetic_batch(2, 1000, seed=6, pad_tail=1)
    _, cfg, esd, _, _, _ = load_case("d1280_1blk_k2")
    _, refs, text = gu.expert_inputs(2, 2, cfg["C"], seed=7)
    expert = VisionExpertHIP(esd, cfg["heads"], cfg["layers"], DEV)
    base = dict(image=image, before_ids=before, after_ids=after, target_ids=tgt, target_mask=tmask)
    for task, key in ((0, "anomaly_maps"), (1, "oneshot_anomaly_maps")):
        model = MyriadHIP(sd, dict(fixed_stage=1, fixed_taskstage=task, need_backward=False), device=DEV)
        model.train()
        with pytest.raises(KeyError):
            model.forward(dict(base))                                  # no maps, no expert: loud failure
        model.attach_vision_expert(expert)
        s1 = dict(base, expert_text_feats=text, ref_images=refs)
        l1 = float(model.forward(s1)["loss"].detach())
        assert s1["anomaly_maps"].shape == (2, 1, 224, 224) and s1["oneshot_anomaly_maps"].shape == (2, 1, 224, 224)
        (zs, _), (osm, _) = expert.forward(image.to(DEV), text, refs)
        l2 = float(model.forward(dict(base, anomaly_maps=zs, oneshot_anomaly_maps=osm))["loss"].detach())
        assert l1 == l2, (key, l1, l2)


@pytest.mark.parametrize("name", ["d256_2blk", "d1024_1blk"])
def test_text_tower_and_prompt_ensemble_vs_reference_golden(name):
    from myriad_amd.vision_expert import ImageBindTextHIP
    from tests.test_expert_oracle import load_text_case
    g, cfg, sd, ids_n, ids_a = load_text_case(name)
    tt = ImageBindTextHIP(sd, cfg["heads"], DEV)
    en = tt.forward(ids_n)
    # embeddings are unit vectors times 1/0.07 = 14.3: 2e-2 of max-abs like every activation comparison
    assert relerr(en, g["emb_normal"]) < 2e-2
    feats = tt.prompt_ensemble(ids_n, ids_a, cfg["n_obj"])
    assert (feats.cpu() - torch.from_numpy(g["text_feats"])).abs().max() < 4e-3      # unit vectors, C = 1024
    assert feats.shape == (cfg["n_obj"], 2, 1024)
