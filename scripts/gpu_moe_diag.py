import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from oracle import llama_oracle as lo
from painlessinferenceacceleration_amd.llama_engine import LlamaVerifyEngine
from tests.tiny_model import GOLDEN, TINY_MOE, moe_shape, moe_weights
g = np.load(os.path.join(GOLDEN, 'moe_tiny_bf16.npz'))
shape = moe_shape(TINY_MOE)
sd = {k: v.to(torch.bfloat16) for k, v in moe_weights(TINY_MOE, 0, torch.float32).items()}
eng = LlamaVerifyEngine(shape, dict(sd), max_length=256)
oracle = lo.OracleLlama(shape, sd)
ids, ref = g['mixtral_0_ids'], g['mixtral_0_logits']
P = 24
eng.prefill(ids[:P].tolist())
lg, _ = oracle.forward(torch.from_numpy(ids[:P]), torch.tril(torch.ones((P, P), dtype=torch.long)), None)
got = eng.logits()[:P].float().cpu()
err = (got - torch.from_numpy(ref[:P])).abs().max(-1).values
erro = (got - lg.float()).abs().max(-1).values
rw = eng._view(9, 64 * 8 * 4, torch.float32).view(64, 8)[:P].cpu()
for t in range(P):
    line = f'row {t:2d} err_ref {float(err[t]):.3f} err_oracle {float(erro[t]):.3f} |'
    for li, rl in enumerate(oracle.router_trace):
        p = torch.softmax(rl[t].float(), -1)
        v, s = torch.topk(p, 3)
        line += f' L{li} top3 {s.tolist()} p {[round(float(x), 3) for x in v]}'
    sel = [i for i in range(8) if rw[t, i] != 0]
    line += f' | engine L1 sel {sel} w {[round(float(rw[t, i]), 3) for i in sel]}'
    print(line)
