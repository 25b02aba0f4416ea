import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import vorbis_amd
from oracle import ref
import tests.test_reference_matrix as m
data = m.gen_windowed_sine()
ch, rate, q = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
pcm = np.tile(data[None, :], (ch, 1)).astype(np.float32)
got = ref.RefEncoder(ch, rate, q, hybrid=True).encode_stream(pcm, tolerate=True)
print("hybrid errors:", [(k, b["error"], b["W"]) for k, b in enumerate(got) if b["error"]], len(got))
e = ref.RefEncoder(ch, rate, q)
want = e.encode_stream(pcm)
an = vorbis_amd.Analyzer(ref.RefEncoder(ch, rate, q).pack_setup(), 0)
print("qlimit", [an.quant_limit(1, c) for c in range(ch)])
for k, b in enumerate(want):
    W = b["W"]
    o = an.analyze(torch.from_numpy(b["pcm"][None]).cuda(), W=W, lW=b["lW"], nW=b["nW"], blocktype=b["blocktype"], ampmax_in=b["ampmax_in"], want=("iwork", "status", "nonzero"))
    torch.cuda.synchronize()
    st = o["status"].cpu().numpy()[0]; iw = o["iwork"].cpu().numpy()[0]
    t = e.tap_block(b["pcm"], b["lW"], W, b["nW"], b["blocktype"], b["ampmax_in"])
    print(k, "W", W, "status", st.tolist(), "max|iwork| gpu", np.abs(iw).max(axis=1).tolist(), "ref", np.abs(t["iwork"]).max(axis=1).tolist(), an.input_status())
