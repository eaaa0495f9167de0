#!/usr/bin/env python3
"""bench.py's smartstart_vec leg alone (for rocprofv3 --kernel-trace --stats): the vectorised SmartStart step at 65 536 envs."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

bench.bench_smartstart_vec(argparse.Namespace(steps=int(sys.argv[1]) if len(sys.argv) > 1 else 40), torch)
