"""400 training steps of the plugin on a learnable synthetic task (HR = smooth images, LR = their
4x average pooling): the loss must fall and stay finite with every step-level fusion switched on."""
import sys, types, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from larvanet_amd.models import LarvaNet as L
m = L.create_model()
m.parse_args(["--num_modules=4", "--num_blocks=4,4,4,4"])
torch.manual_seed(0)
m.prepare(is_training=True, scales=[4])
m.sync_loss = True
dev = m.device
g = torch.Generator().manual_seed(1)
# a learnable synthetic task: HR = smooth image, LR = its 4x average pooling
hr = torch.nn.functional.interpolate(torch.rand(16, 3, 12, 12, generator=g) * 255, scale_factor=16, mode="bicubic", align_corners=False).clamp(0, 255)
lr = torch.nn.functional.avg_pool2d(hr, 4)
hr, lr = hr.to(dev).contiguous(), lr.to(dev).contiguous()
class Val:
    def get_num_images(self): return 1
    def get_image_pair(self, image_index, scale):
        return lr[0].cpu().numpy(), hr[0].cpu().numpy(), "v"
args = types.SimpleNamespace(train_path="/tmp")
losses = []
for i in range(int(os.environ.get("STEPS", "400"))):
    losses.append(m.train_step_larva(args, Val(), lr, hr))
print("loss step 1 %.4f, 50 %.4f, 100 %.4f, 200 %.4f, last %.4f" % (losses[0], losses[49], losses[99], losses[199], losses[-1]))
assert all(np.isfinite(losses)) and losses[-1] < 0.5 * losses[0], "training does not converge"
print("graph in use:", m.use_hip_graph, " psnr of the fit:", m.validate_for_train(args, Val()))
