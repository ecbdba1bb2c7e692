import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from torch.profiler import profile, ProfilerActivity
from autoprog_amd.models import create_model
from autoprog_amd.loss import TokenLabelCrossEntropy
from autoprog_amd.dist import GradientBucketReducer
torch.backends.cudnn.benchmark = True
model = create_model("model_variant", variant="volo_h12_l18", drop_path_rate=0.1).cuda().train()
loss_fn = TokenLabelCrossEntropy(dense_weight=0.5, cls_weight=1.0)
red = GradientBucketReducer(list(model.parameters()), world_size=1); red.install_sink()
B = 128
x = torch.randn(B, 3, 224, 224, device="cuda"); t = torch.rand(B, 1000, 198, device="cuda")
def step():
    red.zero_grad(); l = loss_fn(model(x), t); l.backward(); red.finish()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60))
