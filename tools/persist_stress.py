import sys, os, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import valle_amd
from bench import S_TEXT, P_PROMPT, synth_inputs
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = valle_amd.VALLE(1024, 16, 12, prefix_mode=1, engine_dtype="bf16").to(dev).eval()
eng = model.engine_for(1, S_TEXT, P_PROMPT)
eng.set_option("ignore_eos", 1)
x, y = synth_inputs(0)
X, Y = x[None].to(dev), y[None].to(dev)
ref = None
t0 = time.time()
n = 0
variants = [(0, 0x33114), (3, 0x2431114), (3, 0x230114), (0, 0x114), (3, 0x1330114), (1, 0x33114), (2, 0x33114)]
while time.time() - t0 < float(sys.argv[1]):
    pf, mode = variants[n % len(variants)]
    eng.set_option("persist", 1); eng.set_option("persist_pf", pf); eng.set_option("persist_mode", mode)
    eng.set_option("trace_ar_logits", 1)
    eng.prefill(X, [S_TEXT], Y, [P_PROMPT])
    codes, gl = eng.generate(top_k=1, max_new=120)
    lg = eng.fetch_ar_logits()[:, 0].clone()
    fail = eng.fetch_u32("persist_fail")
    if ref is None: ref = lg
    ok = torch.equal(ref, lg)
    if not ok or fail:
        print("MISMATCH", n, pf, hex(mode), fail, (ref - lg).abs().max().item(), flush=True)
    n += 1
    if n % 50 == 0: print("iter", n, flush=True)
print("done", n, "decodes")
