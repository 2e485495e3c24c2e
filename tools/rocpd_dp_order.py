#!/usr/bin/env python3
"""Order of events on ONE data-parallel rank's device timeline (rocprofv3 --kernel-trace --memory-copy-trace of tests/dp_worker.py,
tools/r6_dp_timeline.sh): per optimisation step, when the LLaMA backward ends, when the map tokenizer's backward ends, when the
first bytes of the tokenizer segment's collective move (the stand-in RCCL stages its chunks through host memory: device-to-host
copies of 16 MiB), when the Q-Former backward's first kernel starts, when the rest of the buffer is exchanged.
VERDICT r5 item 2: the segment's collective is enqueued before qformer.backward's first kernel.
Usage: python tools/rocpd_dp_order.py <results.db> > profiles/r06_dp_segment_order.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
ktab = "kernels" if "kernels" in tables else next(t for t in tables if "kernel" in t.lower())
mtab = next((t for t in tables if t.lower() in ("memory_copies", "memory_copy")), None) or next(t for t in tables if "memory_cop" in t.lower())
mcols = [r[1] for r in db.execute(f"pragma table_info({mtab})")]
size_col = next(c for c in ("size", "bytes", "size_bytes") if c in mcols)
name_col = next((c for c in ("name", "kind", "direction") if c in mcols), None)
kern = db.execute(f"select name, start, end from {ktab} order by start").fetchall()
cop = db.execute(f"select {name_col or 'null'}, start, end, {size_col} from {mtab} order by start").fetchall()
ce = [s for n, s, e in kern if "clamp_ce" in n]
# a training step = [forward clamp_ce, the next forward clamp_ce); clamp_ce runs twice per step (forward, backward): take every other
steps = list(zip(ce[0::2], ce[2::2] + [kern[-1][2] + 1]))
t0 = kern[0][1]
ms = lambda t: f"{(t - t0) / 1e6:9.3f}"
is_copy = lambda n: "copyBuffer" in n or "rocclr_copy" in n
print("# One data-parallel rank (of two on one GPU, mh_ctx verbs over the stand-in RCCL): order of events per optimisation step\n")
print("Times in ms from the first kernel of the trace.  The stand-in RCCL stages every 16 MiB chunk of a collective through host memory "
      "(hipMemcpy device -> pageable host -> device), which the trace shows as runs of the runtime's copy kernels "
      "(`__amd_rocclr_copyBuffer*`) and / or memory-copy records; `window` = from the end of the LLaMA backward's last attention-backward "
      "kernel to the first kernel of the Q-Former backward.\n")
print("| step | LLaMA backward ends | tokenizer backward's last conv kernel ends | first copy of the early segment | Q-Former backward's "
      "first kernel | copy kernels in the window | memory-copy records in the window (MiB) | copy kernels after the Q-Former backward "
      "started | verdict |")
print("|---|---|---|---|---|---|---|---|---|")
for i, (a, b) in enumerate(steps):
    ks = [(n, s, e) for n, s, e in kern if a <= s < b]
    att = [e for n, s, e in ks if "attn_seq_bwd" in n]
    if not att:
        continue
    t_llm = max(att)
    qf = [s for n, s, e in ks if ("attn_bwd_dq" in n or "attn_bwd_dkv" in n) and s > t_llm]
    if not qf:
        continue
    t_qf = min(qf)
    conv = [e for n, s, e in ks if ("col2im" in n or "relu_pool_bwd" in n) and t_llm < s < t_qf]
    t_tok = max(conv) if conv else t_llm
    win_k = [(n, s, e) for n, s, e in ks if is_copy(n) and t_tok <= s < t_qf]
    win_c = [c for c in cop if t_tok <= c[1] < t_qf]
    late_k = [(n, s, e) for n, s, e in ks if is_copy(n) and s >= t_qf]
    first = min([s for _, s, _ in win_k] + [c[1] for c in win_c], default=None)
    early = len(win_k) >= 8 or sum(c[3] for c in win_c) >= (64 << 20)
    verdict = "the segment's collective ran BEFORE the Q-Former backward started" if early else "nothing exchanged before the Q-Former backward"
    print(f"| {i} | {ms(t_llm)} | {ms(t_tok) if conv else '(tokenizer unused at this stage)'} | {ms(first) if first else '-'} | {ms(t_qf)} | "
          f"{len(win_k)} | {sum(c[3] for c in win_c) / 2 ** 20:.0f} | {len(late_k)} | {verdict} |")
