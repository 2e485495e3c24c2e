import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from myriad_amd.llama import LlamaHIP
from myriad_amd import ops
from tests import golden_utils as gu
ops.ensure_workspace("cuda:0")
sd = gu.llama_weights(4096, 2, 11008, 32000, seed=401)
lm = LlamaHIP(sd, 32, "cuda:0", need_backward=False)
emb = (torch.randn(2, 143, 4096) * 0.02).cuda()
for ug in (False, True):
    print("use_graph", ug, flush=True)
    ids = lm.greedy_generate(emb, max_new_tokens=12, stop_ids=((-1,),), use_graph=ug)
    torch.cuda.synchronize()
    print(ids, flush=True)
