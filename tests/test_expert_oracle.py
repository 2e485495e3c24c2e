"""the oracle's restatement of the vision expert (ImageBind vision trunk + anomaly-map heads) against
goldens produced by the reference's own ImageBindModel class and adrefexpert.forward code (tools/make_golden_expert.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import expert_ref as X
from tests import golden_utils as gu

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load_case(name):
    g = np.load(os.path.join(GOLD, f"expert_{name}.npz"))
    D, heads, blocks, C, B, k, seed = (int(v) for v in g["cfg"])
    layers = [int(v) for v in g["layers"]]
    sd = gu.expert_weights(D, blocks, C, len(layers), seed)
    images, refs, text = gu.expert_inputs(B, k, C, seed + 100)
    return g, dict(D=D, heads=heads, blocks=blocks, C=C, B=B, k=k, layers=layers), sd, images, refs, text


@pytest.mark.parametrize("name", ["d1280_3blk", "d1280_1blk_k2"])
def test_expert_oracle_matches_reference(name):
    torch.set_num_threads(8)
    g, cfg, sd, images, refs, text = load_case(name)
    emb, taps = X.vision_trunk(sd, images, cfg["heads"], cfg["layers"], cfg["blocks"])
    np.testing.assert_allclose(emb.numpy(), g["image_embeds"], atol=2e-5)
    for i, t in enumerate(taps):
        np.testing.assert_allclose(t[:, ::8, ::16].numpy(), g[f"tap{i}_sub"], atol=2e-4, rtol=1e-4)
    dec = {k[len("image_decoder."):]: v for k, v in sd.items() if k.startswith("image_decoder.")}
    zmap, zmask = X.zero_shot_maps(taps, dec, text)
    np.testing.assert_allclose(zmap.numpy(), g["zs_map"], atol=2e-4)
    np.testing.assert_allclose(zmask.numpy(), g["zs_mask"], atol=2e-4)
    _, rtaps = X.vision_trunk(sd, refs, cfg["heads"], cfg["layers"], cfg["blocks"])
    omap, omask = X.one_shot_maps(taps, rtaps)
    np.testing.assert_allclose(omap.numpy(), g["os_map"], atol=2e-5)
    np.testing.assert_allclose(omask.numpy(), g["os_mask"], atol=2e-5)
    assert zmap.shape == (cfg["B"], 1, 224, 224) and omask.shape == (cfg["B"], 1, 16, 16)


def load_text_case(name):
    g = np.load(os.path.join(GOLD, f"expert_text_{name}.npz"))
    Dt, heads, blocks, C, n_obj, seed = (int(v) for v in g["cfg"])
    sd = gu.expert_text_weights(Dt, blocks, C, seed)
    ids_n, ids_a = gu.expert_prompt_ids(n_obj, 14, 10, seed + 1)
    return g, dict(Dt=Dt, heads=heads, blocks=blocks, C=C, n_obj=n_obj), sd, ids_n, ids_a


@pytest.mark.parametrize("name", ["d256_2blk", "d1024_1blk"])
def test_text_tower_and_prompt_ensemble_match_reference(name):
    torch.set_num_threads(8)
    g, cfg, sd, ids_n, ids_a = load_text_case(name)
    en = X.text_trunk(sd, ids_n, cfg["heads"], cfg["blocks"])
    np.testing.assert_allclose(en.numpy(), g["emb_normal"], atol=3e-5, rtol=1e-5)         # scaled by 1/0.07 = 14.3
    ea = X.text_trunk(sd, ids_a, cfg["heads"], cfg["blocks"])
    feats = X.text_prompt_ensemble(en, ea, cfg["n_obj"])
    np.testing.assert_allclose(feats.numpy(), g["text_feats"], atol=2e-6)
    assert feats.shape == (cfg["n_obj"], 2, 1024)
