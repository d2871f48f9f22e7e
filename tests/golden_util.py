"""Helpers shared by the CPU (oracle) and GPU (HIP) golden-vector tests."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def conv_i8_fixtures():
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "conv_i8_*.npz")))


def conv_f32_fixtures():
    return ["conv_f32_3x3", "conv_f32_1x1s2", "conv_f32_7x7s2"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))
