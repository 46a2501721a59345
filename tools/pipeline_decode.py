"""Does the ODE of batch i+1 hide inside the SD-VAE decode of batch i?  (two streams, under AMD_DIRECT_DISPATCH=0)
Serial: sampler then decode, batch after batch.  Pipelined: sampler on stream A, decode on stream B behind an event."""
import os, sys, time
os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from flocoder_amd.codecs import SD_VAE_Wrapper
from flocoder_amd.sampling import decode_latents, euler_sampler
dev = torch.device("cuda", 0)
model = bench.build_model(dev)
vae = SD_VAE_Wrapper(weights="random", seed=0).eval().to(dev)
if len(sys.argv) > 1: vae.set_precision(sys.argv[1])
N, B = 6, 64
g = torch.Generator().manual_seed(1)
noise = [torch.randn(B, 4, 32, 32, generator=g).to(dev) for _ in range(N)]
ids = [torch.randint(102, (B,), generator=g).to(dev) for _ in range(N)]
shape = (B, 4, 32, 32)

def serial():
    outs = []
    for i in range(N):
        lat = euler_sampler(model, shape, 64, cond=ids[i], source=noise[i])[0]
        outs.append(decode_latents(vae, lat * 4.5, chunk_size=16))
    return outs

def pipelined():
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    cur = torch.cuda.current_stream(dev)
    sa.wait_stream(cur); sb.wait_stream(cur)
    outs, evs, lats = [], [], []
    model.set_shared_device(True)
    try:
        for i in range(N):
            with torch.cuda.stream(sa):
                lat = euler_sampler(model, shape, 64, cond=ids[i], source=noise[i])[0] * 4.5
                ev = torch.cuda.Event(); ev.record(sa)
            lats.append(lat)
            with torch.cuda.stream(sb):
                sb.wait_event(ev)
                outs.append(decode_latents(vae, lat, chunk_size=16))
        cur.wait_stream(sa); cur.wait_stream(sb)
    finally:
        model.set_shared_device(None)
    return outs

for name, fn in (("serial", serial), ("pipelined", pipelined), ("serial", serial), ("pipelined", pipelined)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); o = fn(); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("%-10s %.1f ms for %d batches = %.1f decoded images/s" % (name, t * 1e3, N, N * B / t), flush=True)
    if name == "serial": ref = [x.clone() for x in o]
    else: print("   max rel diff vs serial %.2e" % max(float((a - b).norm() / b.norm()) for a, b in zip(o, ref)))
