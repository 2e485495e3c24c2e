"""One data-parallel rank of tests/test_dp_gpu.py: `python tests/dp_worker.py RANK WORLD PORT MODE OUT [COLLECTIVE WIRE RCCL_LIB]`.
Every rank sits on cuda:0 (MYRIAD_SINGLE_DEVICE=1) and the process group is gloo -- RCCL refuses two ranks per device -- so this
is the N > 1 control flow of runner.DataParallel + MyriadHIP.train_step (side-stream exchange, delayed AdamW under the next step's
ViT, use flags riding the buffer, rs_ag shards), not a measurement of the wire.  COLLECTIVE = torch: the data goes through
torch.distributed (gloo); ctx: through the C ABI's mh_ctx verbs (csrc/ctx.hip: events, side stream, the dlsym'd ncclAllReduce /
ReduceScatter / AllGather argument lists -- the default path on an 8-GPU node) bound to the stand-in of tests/fake_rccl via
MYRIAD_RCCL_LIB, gloo then only carries the communicator id; ctx_bad0: the same with rank 0 unable to load its RCCL (all ranks
must agree to fall back).  WIRE = f32 | bf16 (MYRIAD_DP_GRAD_DTYPE)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, mode, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    collective = sys.argv[6] if len(sys.argv) > 6 else "torch"
    wire = sys.argv[7] if len(sys.argv) > 7 else "f32"
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                      MYRIAD_DIST_BACKEND="gloo", MYRIAD_SINGLE_DEVICE="1",
                      MYRIAD_DP_COLLECTIVE="ctx" if collective.startswith("ctx") else collective, MYRIAD_DP_GRAD_DTYPE=wire)
    if collective.startswith("ctx"):
        # ctx_bad0: rank 0 alone cannot load its RCCL -- every rank must then fall back to torch.distributed together
        os.environ["MYRIAD_RCCL_LIB"] = "/nonexistent/librccl.so" if (collective == "ctx_bad0" and rank == 0) else sys.argv[8]
    import torch
    from myriad_amd.runner import DataParallel, init_distributed
    from tests import dp_common as C
    r, w, local = init_distributed()
    assert (r, w, local) == (rank, world, 0)
    dev = torch.device("cuda:0")
    model, cfg = C.build_model(dev)
    model.lora.base_seed = C.BASE_SEED + rank                    # seed + rank (train.py:63-72)
    dp = DataParallel(dev, mode=mode)
    assert dp.world == world and dp.side is not None
    assert (dp.ctx is not None) == (collective == "ctx"), "the requested exchange path did not come up (or a broken one did)"
    assert dp.grad_dtype == (torch.bfloat16 if wire == "bf16" else torch.float32)
    batches = [C.batch(rank, i, cfg["vocab"], dev) for i in range(C.N_STEPS)]
    losses = []
    for i in range(C.N_STEPS):
        model.fixed_stage = C.STAGES[rank][i]
        nxt = batches[i + 1] if i + 1 < C.N_STEPS else None
        losses.append(model.train_step(batches[i], C.LRS[i], 0.05, dp=dp, world=world, overlap=True, next_samples=nxt))
    model.finish_update()
    if mode == "rs_ag":
        dp.gather_state(model.store)
    snap = C.snapshot(model)
    snap["losses"] = [float(l) for l in losses]
    torch.save(snap, out)
    dp.barrier()


if __name__ == "__main__":
    main()
