"""Shared by tests/test_dp_gpu.py and its worker processes (tests/dp_worker.py): a reduced-depth, full-width Myriad model,
per-rank batches and the per-rank prompt stages of a three-step data-parallel run."""
import torch

# prompt stage per rank per step (reference myriad.py:378 draws it per forward from the rank's own RNG, so ranks disagree):
#   step 0: rank 1 skips VETokenizer (stage 2) while rank 0 uses everything      -> ragged use, flags summed by the exchange
#   step 1: every rank at stage 0 -> VEInstructor unused everywhere               -> its weights, moments and step count stay
#   step 2: rank 0 skips VETokenizer, rank 1 uses everything
STAGES = {0: [1, 0, 2], 1: [2, 0, 1]}
N_STEPS = 3
LRS = [1e-3, 8e-4, 6e-4]
BASE_SEED = 42


def build_model(dev):
    from myriad_amd.myriad import MyriadHIP
    from myriad_amd.synthetic import SyntheticWeights, full_config
    cfg = full_config(vit_depth=1, qf_layers=2, llm_layers=2, vocab=1024)
    torch.manual_seed(1234)                         # trainable init (identical on every rank, as DDP broadcasts rank 0's)
    w = SyntheticWeights(cfg, dev, seed=7)
    return MyriadHIP(w, dict(fixed_stage=1, fixed_taskstage=0, use_lora=True), device=dev), cfg


def batch(rank, step, vocab, dev):
    g = torch.Generator().manual_seed(1000 + 37 * rank + step)
    B = 2
    image = torch.randn(B, 3, 224, 224, generator=g)
    maps = torch.rand(B, 1, 224, 224, generator=g)
    before = torch.randint(3, vocab, (1, 4), generator=g).expand(B, -1).contiguous()
    after = torch.randint(3, vocab, (1, 28), generator=g).expand(B, -1).contiguous()
    tgt = torch.randint(3, vocab, (B, 16), generator=g)
    return dict(image=image.to(dev), anomaly_maps=maps.to(dev), oneshot_anomaly_maps=maps.to(dev), before_ids=before,
                after_ids=after, target_ids=tgt, target_mask=torch.ones(B, 16, dtype=torch.long))


def snapshot(model):
    st = model.store
    torch.cuda.synchronize()
    return dict(p=st.flat_p.cpu().clone(), m=st.flat_m.cpu().clone(), v=st.flat_v.cpu().clone(), steps=st.module_steps())
