import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flocoder_amd import _binding as B
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
noise, ids = bench.synthetic_inputs(0, 1, dev)
t = torch.full((bench.BATCH,), 500.0, device=dev)
buf = torch.zeros(2 * 8192 * 8 * 16, dtype=torch.int64, device=dev)
with torch.no_grad():
    for _ in range(3): model(noise, t, {"class_cond": ids})
    B.check(B.lib().fc_debug_set_conv_stamps(buf.data_ptr()))
    model(noise, t, {"class_cond": ids})
    B.check(B.lib().fc_debug_set_conv_stamps(None))
torch.cuda.synchronize()
st = buf[8192 * 8 * 16:].view(-1, 4, 16)[:256].cpu().double()   # [block][wave][16]
names = ["start", "stats", "tables", "x staged", "gemm done", "attn done", "end"]
for w in range(4):
    rel = st[:, w, :7] - st[:, w, 0:1]
    print("wave", w, ", ".join(f"{n}={int(v)}" for n, v in zip(names, rel.median(dim=0).values.tolist())))
print("units: s_memtime ticks (100 MHz => 10 ns)" )
