"""Host-side mirror of the reference's `diff_surfel_tracing` interface (easyvolcap/utils/optix_utils.py:7,24,78,104-119,188-201).
Filled in with the HIP LBVH tracer; this first revision only carries the interface records."""
from typing import NamedTuple

import torch
from torch import nn


class SurfelTracingSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool
    max_trace_depth: int
    specular_threshold: float


class SurfelTracer(nn.Module):
    def __init__(self):
        super().__init__()

    def build_acceleration_structure(self, vertices, faces, rebuild=True):
        raise RuntimeError("envgs_amd: the HIP LBVH tracer is not built into this revision")

    def forward(self, *a, **k):
        raise RuntimeError("envgs_amd: the HIP LBVH tracer is not built into this revision")
