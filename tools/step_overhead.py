import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from tokenizer_amd import capi
from tools import corpora
ctx = capi.Context(0)
n_docs = 65536
text, off = corpora.english(n_docs, 1024)
d_text = torch.from_numpy(text).cuda(); d_off = torch.from_numpy(off.astype(np.int64)).cuda()
cap = text.size // 2 + 2 * n_docs
d_out = torch.empty(cap, dtype=torch.int32, device="cuda"); d_oo = torch.empty(n_docs + 1, dtype=torch.int64, device="cuda")
for prof in (False, True, False):
    ctx.set_profiling(prof)
    for _ in range(5):
        ctx.encode_batch_device(d_text.data_ptr(), text.size, d_off.data_ptr(), n_docs, 3, d_out.data_ptr(), cap, d_oo.data_ptr(), None)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        ctx.encode_batch_device(d_text.data_ptr(), text.size, d_off.data_ptr(), n_docs, 3, d_out.data_ptr(), cap, d_oo.data_ptr(), None)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("profiling", prof, "ms/step %.4f" % (dt * 1e3), flush=True)
