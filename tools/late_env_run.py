#!/usr/bin/env python
"""The faithful run of a bench workload in a process whose HIP runtime initialised BEFORE GPU_MAX_HW_QUEUES reached the
environment (torch touched the GPU first): the runtime then has its default 4 hardware queues, the pipeline's streams share
them, and the engine falls back to as many slots as its probe measures to run side by side (engine.hip, spec_ensure).
usage: IPC_SPEC_STATS=1 python tools/late_env_run.py C1"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.pop("GPU_MAX_HW_QUEUES", None)
import torch  # noqa: E402

torch.cuda.init()
torch.zeros(1, device="cuda")          # HIP initialises with the runtime's default number of hardware queues
sys.argv = ["lib_incremental.py", os.path.join(ROOT, "ipc_amd", "libipc_amd.so"), sys.argv[1] if len(sys.argv) > 1 else "C1", "1"]
runpy.run_path(os.path.join(ROOT, "tools", "lib_incremental.py"), run_name="__main__")
