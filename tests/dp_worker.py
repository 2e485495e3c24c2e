"""One data-parallel rank of tests/test_dp_gpu.py: `python tests/dp_worker.py RANK WORLD PORT MODE OUT`.  Every rank sits on
cuda:0 (MYRIAD_SINGLE_DEVICE=1) and the process group is gloo -- RCCL refuses two ranks per device -- so this is the N > 1
control flow of runner.DataParallel + MyriadHIP.train_step (side-stream exchange, delayed AdamW under the next step's ViT,
use flags riding the buffer, rs_ag shards), not a measurement of the wire."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, mode, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                      MYRIAD_DIST_BACKEND="gloo", MYRIAD_SINGLE_DEVICE="1")
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
