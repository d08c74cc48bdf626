import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "eta-inversion_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from run_configs import run
run("2: etainv+simple 512^2 fp16 S=50 B=1", 64, 50, 1, "simple", torch.float16, reps=1)
