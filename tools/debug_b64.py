import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import valle_amd
from bench import P_PROMPT, S_TEXT, synth_inputs
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
same = len(sys.argv) > 2 and sys.argv[2] == "same"
model = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16", max_batch=B).to(dev).eval()
def run(Bn, idxs):
    eng = model.engine_for(Bn, S_TEXT, P_PROMPT)
    eng.set_option("trace_ar_logits", 1)
    X = torch.stack([synth_inputs(i)[0] for i in idxs]).to(dev)
    Y = torch.stack([synth_inputs(i)[1] for i in idxs]).to(dev)
    eng.prefill(X, [S_TEXT] * Bn, Y, [P_PROMPT] * Bn)
    try:
        codes0, gl = eng.generate(top_k=1, max_new=4)
    except Exception as e:
        print("generate error:", e); gl = eng._gen_lens
    lg = eng.fetch_ar_logits()   # [steps+1, B, 1025]
    return lg, gl
idxs = [0] * B if same else list(range(B))
lgB, glB = run(B, idxs)
print("B =", B, "gen_lens", glB[:16], "...")
am = lgB[0].argmax(-1)
print("step0 argmax per utt:", am.tolist())
print("step0 EOS logit rank: ", [(int((lgB[0, b] > lgB[0, b, 1024]).sum())) for b in range(min(B, 16))])
print("nan/inf rows:", [b for b in range(B) if not torch.isfinite(lgB[0, b]).all()])
# reference: each utterance alone
bad = []
for b in list(range(min(B, 6))) + [B - 1]:
    lg1, gl1 = run(1, [idxs[b]])
    d = (lg1[0, 0] - lgB[0, b]).abs().max().item()
    print(f"utt {b}: max|dlogit| step0 vs B=1: {d:.4f}  (sigma {lg1[0,0].std().item():.3f}) argmax {int(lg1[0,0].argmax())} vs {int(lgB[0,b].argmax())}")
