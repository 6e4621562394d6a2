#!/usr/bin/env python3
"""cfg4 only (6 modes x Fock cutoff 32): quick timing of the squeezing / beam-splitter kernels."""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
from bench_configs import cv_fock
print(json.dumps(cv_fock()))
