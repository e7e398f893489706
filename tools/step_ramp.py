import sys, time, types, importlib, torch
sys.path.insert(0, '/root/repo')
import bench
dev = torch.device('cuda', 0)
m = importlib.import_module('larvanet_amd.models.LarvaNet').create_model()
m.parse_args(list(bench.FLAGS)); torch.manual_seed(0)
m.volume_per_step = 48*48*16*3
m.prepare(is_training=True, scales=[4]); m.sync_loss = True
g = torch.Generator().manual_seed(1000)
x = (torch.rand(16,3,48,48, generator=g)*255).to(dev); t = (torch.rand(16,3,192,192, generator=g)*255).to(dev)
args = types.SimpleNamespace(train_path='/tmp'); val = bench.TinyValLoader()
for _ in range(5): m.train_step_larva(args, val, x, t)
torch.cuda.synchronize()
out = []
for blk in range(16):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.train_step_larva(args, val, x, t)
    torch.cuda.synchronize(); out.append((time.perf_counter()-t0)/10*1e3)
print('ms per step in consecutive blocks of 10 steps after 5 warm-up steps:', ' '.join('%.4f' % v for v in out))
time.sleep(0.5)
out = []
for blk in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m.train_step_larva(args, val, x, t)
    torch.cuda.synchronize(); out.append((time.perf_counter()-t0)/10*1e3)
print('after a 0.5 s pause:', ' '.join('%.4f' % v for v in out))
