#!/usr/bin/env python3
"""Diagnostic: ray-cast replay time per scan for a given library build. usage: time_raycast.py [libname]"""
import os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "iterative-closest-point-avmi_amd")); sys.path.insert(0, REPO)
from icpmi import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
import numpy as np, torch
from icpmi import synth
import bench
r = bench.bench_raycast(torch, synth, 200, False)
print(os.path.basename(_lib.LIB_PATH), "ms_per_scan", r["ms_per_scan"], "device", r["device_ms_per_scan"], "cells/s %.3e" % r["cells_per_sec"], "single scan us", r["single_scan"]["us_per_scan"])
