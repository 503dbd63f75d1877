import os, sys
sys.path.insert(0, "tests")
import torch
print("device_count", torch.cuda.device_count())
os.environ["GPU_MAX_HW_QUEUES"] = "16"          # AFTER import torch and device_count(), before any GPU call
sys.argv = ["_ingraph_worker.py", "4", "0", "1500"]
import _ingraph_worker
_ingraph_worker.main()
