"""FFT-faithful torch-CPU restatement of the reference Taxim path (TEST / BASELINE INFRASTRUCTURE).

This is the "reference CPU path" leg of bench.py (`cpu_baseline.kind == "port"`): the same algorithm and the
same tensor library the reference runs on a CPU - reflect-pad + torch.fft cross-correlation x 7, masked
restore x 6, normals, 125x125x6 polynomial gather, feature dot product, + background, clip
(gpu_taxim/sim/taxim_torch.py:19-44,225-258,381-412,443-503) - multi-threaded through torch's intra-op pool.
The reference itself cannot travel to the GPU box; this port is validated against it in
tests/test_oracle_golden.py::test_torch_cpu_port_vs_reference.  Never imported by the product path.
"""
from __future__ import annotations

import math
from pathlib import Path

import numpy as np
import torch

from .taxim_oracle import TaximOracle, gaussian_kernel1d, gaussian_kernel_size


class TaximTorchCpuPort:
    def __init__(self, calib_dir: Path, shape_hw=(240, 320)):
        o = TaximOracle(calib_dir, shape_hw, "direct")  # table preparation only
        self.H, self.W = shape_hw
        self.pixmm = o.p.pixmm
        self.cs = o.p.sim["contact_scale"]
        self.calib_h, self.calib_w = o.p.calib_h, o.p.calib_w
        self.poly = torch.from_numpy(o.poly)
        self.gel = torch.from_numpy(o.gel)
        self.bg = torch.from_numpy(o.bg)
        self.feat = torch.from_numpy(o.feat.reshape(-1, 6))
        self.kernels = []
        for sg in o.pyr_sigmas + [o.final_sigma]:
            kw, kh = gaussian_kernel_size(sg[0]), gaussian_kernel_size(sg[1])
            k2d = torch.from_numpy(gaussian_kernel1d(sg[1], kh))[:, None] @ torch.from_numpy(gaussian_kernel1d(sg[0], kw))[None, :]
            self.kernels.append((kw, kh, k2d))
        nb = o.p.num_bins
        self.x_binr = 0.5 * math.pi / (nb - 1)
        self.y_binr = 2 * math.pi / (nb - 1)

    @staticmethod
    def _fft_corr(x: torch.Tensor, k2d: torch.Tensor) -> torch.Tensor:
        kh, kw = k2d.shape
        xp = torch.nn.functional.pad(x[None], ((kw - 1) // 2,) * 2 + ((kh - 1) // 2,) * 2, mode="reflect")[0]
        kp = torch.zeros(xp.shape[-2:], dtype=x.dtype)
        kp[:kh, :kw] = k2d
        out = torch.fft.ifft2(torch.fft.fft2(xp) * torch.conj(torch.fft.fft2(kp[None]))).real
        return out[..., : xp.shape[-2] - (kh - 1), : xp.shape[-1] - (kw - 1)]

    @torch.no_grad()
    def render_direct(self, hm: torch.Tensor, press: torch.Tensor) -> torch.Tensor:
        S = hm - hm.amin(-1, keepdim=True).amin(-2, keepdim=True) - press.view(-1, 1, 1)
        P = -S.amin(-1).amin(-1)
        J = torch.minimum(S, self.gel)
        M = torch.logical_and(J - self.gel < -P[:, None, None] * self.cs, S < 0)
        Z = J
        for (kw, kh, k2d) in self.kernels[:-1]:
            Z = self._fft_corr(Z, k2d)
            Z[M] = J[M]
        Z = self._fft_corr(Z, self.kernels[-1][2])
        z = -(Z / self.pixmm)
        h, w = z.shape[-2:]
        dzdx = (z[..., 2:h, 1 : w - 1] - z[..., 0 : h - 2, 1 : w - 1]) / 2.0 * h / self.calib_h
        dzdy = (z[..., 1 : h - 1, 2:w] - z[..., 1 : h - 1, 0 : w - 2]) / 2.0 * w / self.calib_w
        t = torch.sqrt(dzdx**2 + dzdy**2)
        mag = torch.arctan(t)
        valid = t != 0
        dr = torch.zeros_like(t)
        dr[valid] = torch.arctan2(dzdx[valid] / t[valid], dzdy[valid] / t[valid])
        mag = torch.nn.functional.pad(mag, (1, 1, 1, 1), "replicate")
        dr = torch.nn.functional.pad(dr, (1, 1, 1, 1), "replicate")
        im = torch.floor(mag / self.x_binr).long()
        idd = torch.floor((dr + math.pi) / self.y_binr).long()
        params = self.poly[:, im, idd].transpose(0, 1).flatten(-3, -2)
        img = (self.feat.unsqueeze(0) * params).sum(-1).unflatten(-1, (h, w))
        return torch.clip(img + self.bg, 0, 1).movedim(1, 3)
