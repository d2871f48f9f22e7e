"""Streams for serving: several independent batches in flight on one MI355X, none of them sharing a hardware queue.

Host-side mirror of `saber_hip_serving_streams` (include/saber_hip.h, anakin_amd/csrc/api_streams.hip): the HIP runtime serves every
stream of a process from four hardware queues assigned by creation order, so "k nets on k streams" measured anything from 43k to 63k
images/s for the same three ResNet50 INT8 batch-8 nets depending on what else had created streams before
(profiles/r06/multi_stream_curve.txt). The library finds, once per device, up to four streams whose spin kernels overlap pairwise.

Role in the reference: framework/core/worker.h runs one Net per pool thread, each on its own Context<T> streams (saber/core/context.h)."""
import ctypes as C

import torch

from . import lib as L


def serving_streams(n):
    """n torch streams for n nets in flight (round-robin over the device's distinct-queue set), and the size of that set"""
    out = (C.c_void_p * max(n, 1))()
    distinct = C.c_int(0)
    L.check(L.load().saber_hip_serving_streams(n, out, C.byref(distinct)))
    dev = torch.cuda.current_device()
    return [torch.cuda.ExternalStream(int(out[i]), device=dev) for i in range(n)], int(distinct.value)
