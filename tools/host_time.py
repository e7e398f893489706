"""Host-side cost of one train_step_larva call (graph replay + optimizer + bookkeeping), measured with the
GPU idle so that nothing queues: must stay well under the 1.8 ms the GPU needs per step."""
import sys, time, types, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from larvanet_amd.models import LarvaNet as L
m = L.create_model(); m.parse_args(["--num_modules=4", "--num_blocks=4,4,4,4"]); torch.manual_seed(0)
m.prepare(is_training=True, scales=[4]); m.sync_loss = False
dev = m.device
x = (torch.rand(16, 3, 48, 48) * 255).to(dev); t = (torch.rand(16, 3, 192, 192) * 255).to(dev)
class Val:
    def get_num_images(self): return 1
    def get_image_pair(self, image_index, scale): return x[0].cpu().numpy(), t[0].cpu().numpy(), "v"
args = types.SimpleNamespace(train_path="/tmp")
for _ in range(5): m.train_step_larva(args, Val(), x, t)
torch.cuda.synchronize()
# host-only cost: time the calls while the GPU is idle-ahead (sync first, then issue 1 step, measure call return)
ts = []
for _ in range(50):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); m.train_step_larva(args, Val(), x, t); ts.append(time.perf_counter() - t0)
ts.sort(); print("host time per train_step_larva call: median %.0f us, p90 %.0f us" % (ts[25] * 1e6, ts[45] * 1e6))
