# Cluster topology: one "host:port" per task; the position in the list is the task_index.
# Same two names as the reference's settings.py.  The default is a single-host
# 1 PS + 2 workers layout (README scenario 3); edit per experiment, or override without
# editing through DTF_PS_SVRS / DTF_WORKER_SVRS (comma-separated), which the tests use.
import os

ps_svrs = ["localhost:2223"]
worker_svrs = ["localhost:2222", "localhost:2221"]

if os.environ.get("DTF_PS_SVRS") is not None:
    ps_svrs = [a for a in os.environ["DTF_PS_SVRS"].split(",") if a]
if os.environ.get("DTF_WORKER_SVRS") is not None:
    worker_svrs = [a for a in os.environ["DTF_WORKER_SVRS"].split(",") if a]
