import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "neo-planner_amd"))
import neo_planner_amd as npa
if sys.argv[1] == "ours_first":
    ctx = npa.Context(0)
    import torch
    try:
        torch.cuda.init(); print("ours first -> torch ok", torch.cuda.device_count())
    except Exception as e:
        print("ours first -> torch FAILED:", str(e)[:80])
else:
    import torch
    torch.cuda.init()
    ctx = npa.Context(0)
    print("torch first -> ours ok")
