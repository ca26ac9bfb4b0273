"""Stand-alone evaluation of the distance encoder's heads (DistBase.build_prototypes,
cfl/models/base.py:43-105) for consumers other than the fused pair kernels: the MrCGAN phase
needs `activations` of the target encoder and `one_prototype_activations` of the source
encoder as generator conditions (cfl/models/cfl.py:731-799).  The encoder is frozen during the
post epochs (only generator / discriminator variables are optimised, cfl/models/cfl.py:1087-1096),
so the heads are unpacked once from the engine's fragment-major theta into the reference's
[D, N] layout and run as weight-normalised FC layers (cfl_conv2d_wn_fwd on a 1x1 image)."""
import numpy as np
import torch

from .. import hipabi as H
from .. import hipgan as G


class FrozenHeads(object):
    def __init__(self, engine, act_type=None):
        self.engine = engine
        self.shape = engine.shape
        self.act_type = act_type if act_type not in (None, 'linear') else None
        self.device = engine.device
        self.refresh()
        self._ws = torch.empty(1024, dtype=torch.float32, device=self.device)

    def refresh(self):
        """(Re)read the encoder variables from the engine."""
        p, pd, _ = H.unpack_theta(self.shape, self.engine.theta)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self.device)
        self.sides = []
        for params in (p, pd if pd is not None else p):
            side = {}
            for head in ('outputs', 'proto'):
                if head + '/W' not in params:
                    continue
                V = dev(params[head + '/W'])
                g = dev(params[head + '/g']) if head + '/g' in params else None
                b = dev(params[head + '/b']) if head + '/b' in params else None
                side[head] = (V, g, b)
            self.sides.append(side)

    def normalize(self, x):
        n = self.engine.norm
        x = x.contiguous()
        if n.mul == 1.0 and n.add == 0.0 and not (n.has_lo or n.has_hi):
            return x
        return G.affine_clip(x, n)

    def _fc(self, x, V, g, b):
        B, D = x.shape
        N = V.shape[1]
        d = H.CflConv(B, 1, 1, D, N, 1, 1, 1, 0)
        need = (G.conv_ws_bytes(d) + 3) // 4
        if self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.float32, device=self.device)
        y = torch.empty(B, N, dtype=torch.float32, device=self.device)
        if g is None:   # plain FC (cfl/models/dist.py): unit gains would renormalise; not used by CFL
            raise H.CflHipError('FrozenHeads expects weight-normalised heads')
        G.conv_fwd(d, x, V, g, b, y, self._ws)
        return G.act_fwd(y, self.act_type) if self.act_type else y

    def activations(self, x, side):
        """act(outputs head) [n, L]; side 0 = source encoder, 1 = target encoder; x = RAW rows."""
        V, g, b = self.sides[side]['outputs']
        return self._fc(self.normalize(x), V, g, b)

    def prototype_activations(self, x, side):
        """act(prototype_outputs head) reshaped [n, K, L]."""
        V, g, b = self.sides[side]['proto']
        y = self._fc(self.normalize(x), V, g, b)
        return y.view(x.shape[0], self.shape.K, self.shape.L)
