import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_kernels_gpu as T
from distributed_tensorflow_amd.models.mlp import MLPEngine
gpu = T._engine(T.DEV, with_grads=False); cpu = T._engine("cpu", with_grads=False)
print("B,K,H,C", gpu.B, gpu.K, gpu.H, gpu.C, "lr", float(gpu.lr))
w2_init = gpu.p["W2"].clone(); b2_init = gpu.p["b2"].clone()
gpu.run_chunk(1, use_graph=False); cpu.run_chunk(1, use_graph=False)
torch.cuda.synchronize()
sh = gpu.shadow.cpu()
print("shadow[:6]", sh[:6].tolist())
print("W2init[:6]", w2_init.flatten()[:6].tolist())
print("shadow-W2init max", (sh[:gpu.H*gpu.C] - w2_init.flatten().cpu()).abs().max().item(), "nonzero", int((sh != 0).sum()))
print("zws nonzero halves", int((gpu.zws[:2048] != 0).sum()), int((gpu.zws[2048:] != 0).sum()))
for k in MLPEngine.PARAM_NAMES:
    print(k, (gpu.p[k].cpu() - cpu.p[k]).abs().max().item())
