"""PairModel: what cfl.models.dist.Dist and cfl.models.cfl.CFL share.

Plays the role of the reference's ModelBase / DistBase (cfl/models/base.py:6-146),
but instead of emitting TensorFlow graph nodes it owns a PairEngine: the heads of
``build_prototypes`` and the distances of ``build_dist`` execute inside the fused
HIP kernels (csrc/cfl_hip.hip), selected by (dist_type, weight_norm, has_bias,
act_type).  Variables are addressed by the reference's TensorFlow names
 in checkpoints.
"""
import os

import numpy as np
import torch

from .. import hipabi as H
from ..engine import PairEngine


def xavier_uniform(rng, fan_in, fan_out):
    """tf.contrib.layers.xavier_initializer(): U(+-sqrt(6 / (fan_in + fan_out)))."""
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


class PairModel(object):
    # internal key -> (TF scope fragment, variable leaf names W / b / g)
    HEAD_SCOPES = {}
    MODEL_SCOPE = ''
    ENCODER_SCOPES = ('',)

    def _setup_engine(self, input_size, latent_size, num_components, dist_type, weight_norm,
                      has_bias, act_type, directed, norm, loss, lr, beta1, beta2, batch_size,
                      seed, device):
        self.input_size = int(input_size)
        self.padded_size = (self.input_size + 63) // 64 * 64
        self.device = torch.device(device if device is not None else 'cuda')
        if norm is not None and self.padded_size != self.input_size:
            # zero-padded feature columns must stay zero through a normaliser with a shift / lower clip
            norm = H.make_norm(norm.mul, norm.add, norm.lo if norm.has_lo else None,
                               norm.hi if norm.has_hi else None, valid_cols=self.input_size)
        rng = np.random.RandomState(seed)
        n_enc = 2 if directed else 1
        params = [self._init_params(rng, dist_type, weight_norm, has_bias, latent_size,
                                    num_components) for _ in range(n_enc)]
        self.engine = PairEngine(
            self.padded_size, latent_size, num_components, dist_type, weight_norm, has_bias,
            act_type, directed, norm=norm, loss=loss, lr=lr, beta1=beta1, beta2=beta2,
            device=self.device, params=params[0], params_dst=params[1] if directed else None,
            batch_size=batch_size)

    def _init_params(self, rng, dist_type, weight_norm, has_bias, L, K):
        D, Dp = self.input_size, self.padded_size

        def head(n):
            w = np.zeros((Dp, n), np.float32)
            w[:D] = xavier_uniform(rng, D, n)
            return w
        p = {'outputs/W': head(L)}
        heads = ['outputs']
        if dist_type != 'siamese':
            p['proto/W'] = head(L * K)
            heads.append('proto')
        for h in heads:
            n = p[h + '/W'].shape[1]
            if has_bias:
                p[h + '/b'] = np.zeros(n, np.float32)
            if weight_norm:
                p[h + '/g'] = np.ones(n, np.float32)
        if dist_type == 'monomer':
            p['mono/W'] = xavier_uniform(rng, L, K)
            if weight_norm:
                p['mono/g'] = np.ones(K, np.float32)
        return p

    # -- data plumbing ------------------------------------------------------
    def upload(self, x):
        """host array -> fp32 device tensor through the model's PinnedUploader (asynchronous; CFL_SYNC_UPLOAD=1: the
        synchronous pageable copy)"""
        dev = torch.device(self.device)
        if dev.type != 'cuda' or os.environ.get('CFL_SYNC_UPLOAD', '0') not in ('0', ''):
            return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)
        up = getattr(self, '_uploader', None)
        if up is None:
            from ..input_data import PinnedUploader
            up = self._uploader = PinnedUploader(dev)
        return up.upload(x)

    def to_device(self, x):
        """[n, input_size] array or tensor -> contiguous fp32 device tensor padded to
        the kernels' multiple-of-64 feature width."""
        if isinstance(x, torch.Tensor):
            t = x.to(self.device, torch.float32)
        else:
            t = self.upload(x)
        if t.shape[1] != self.padded_size:
            t = torch.nn.functional.pad(t, (0, self.padded_size - t.shape[1]))
        return t.contiguous()

    # -- the two entry points of the hot path ----------------------------------
    @staticmethod
    def is_indexed(batch):
        """a (feature table, IndexStreams) batch of ResidentFeatures.next_indexed"""
        return len(batch) == 2 and isinstance(batch[1], H.IndexStreams)

    def train_step(self, batch):
        """One optimisation step on (pos_src, pos_dst, neg_src, neg_dst) -- dense rows, or the positions of the
        rows in a resident feature table."""
        if self.is_indexed(batch):
            self.engine.step(batch)
            return
        self.engine.step([self.to_device(b) for b in batch])

    def predict(self, src, dst):
        """Scores [n, 1] = max(thr, 1e-6) - dist(src, dst) (``val_s_pos_predicts.outputs``)."""
        s = self.engine.scores(self.to_device(src), self.to_device(dst))
        return s.cpu().numpy().reshape(-1, 1)

    def batch_accuracy(self, batch):
        """s_accuracy of a batch without training on it (the val_s_accuracy fetch)."""
        if self.is_indexed(batch):
            sp = self.engine.scores(batch[0], batch[1].pair(0))
            sn = self.engine.scores(batch[0], batch[1].pair(1))
            return 0.5 * float((sp > 0).float().mean() + (sn <= 0).float().mean())
        sp = self.engine.scores(self.to_device(batch[0]), self.to_device(batch[1]))
        sn = self.engine.scores(self.to_device(batch[2]), self.to_device(batch[3]))
        return 0.5 * float((sp > 0).float().mean() + (sn <= 0).float().mean())

    def scalars(self):
        return self.engine.read_scalars()

    # -- checkpoints -----------------------------------------------------------
    def _tf_names(self):
        names = {}
        for e, scope in enumerate(self.ENCODER_SCOPES):
            for key, (frag, leaves) in self.HEAD_SCOPES.items():
                for leaf_key, leaf in leaves.items():
                    names[(e, key + '/' + leaf_key)] = '{}/{}/{}/{}'.format(
                        self.MODEL_SCOPE, scope, frag, leaf)
        return names

    def _named(self, theta):
        p, pd, thr = H.unpack_theta(self.engine.shape, theta)
        out = {}
        names = self._tf_names()
        for e, params in enumerate((p, pd)):
            if params is None:
                continue
            for k, v in params.items():
                if k.endswith('/W') and not k.startswith('mono'):
                    v = v[:self.input_size]          # drop the zero padding rows
                out[names[(e, k)]] = v
        out[self.MODEL_SCOPE + '/Thresholder/threshold/threshold'] = np.float32(thr)
        return out

    def checkpoint_state(self):
        eng = self.engine
        eng.require_complete_slots()
        state = {'variables': self._named(eng.theta),
                 'adam_m': self._named(eng.m), 'adam_v': self._named(eng.v),
                 'beta1_power': float(eng.beta1_power), 'beta2_power': float(eng.beta2_power),
                 'global_step': eng.global_step, 'name': self.get_name(),
                 # what the TensorFlow-side names of a converted checkpoint depend on (cfl.bin.convert_checkpoint --to-tf):
                 # stated by the model instead of being parsed out of its name
                 'graph': {'own_threshold_optimiser': bool(getattr(self, 'MODEL_SCOPE', '') == 'CFL'
                                                           and not getattr(self, 'use_threshold', True)),
                           'has_gan': getattr(self, 'gan_phase', None) is not None}}
        return state

    def _pack(self, named):
        names = self._tf_names()
        groups = [dict(), dict()]
        for (e, k), n in names.items():
            if n in named:
                v = np.asarray(named[n], np.float32)
                if k.endswith('/W') and not k.startswith('mono'):
                    full = np.zeros((self.padded_size, v.shape[1]), np.float32)
                    full[:self.input_size] = v
                    v = full
                groups[e][k] = v
        thr = float(named[self.MODEL_SCOPE + '/Thresholder/threshold/threshold'])
        directed = bool(self.engine.shape.directed)
        return H.pack_theta(self.engine.shape, groups[0], groups[1] if directed else None, thr,
                            self.device)

    def load_checkpoint_state(self, state):
        eng = self.engine
        eng.set_theta(self._pack(state['variables']))
        eng.m.copy_(self._pack(state['adam_m']))
        eng.v.copy_(self._pack(state['adam_v']))
        eng.beta1_power = np.float32(state['beta1_power'])
        eng.beta2_power = np.float32(state['beta2_power'])
        eng.global_step = int(state['global_step'])

    def assign_trainable(self, state, ignore_missing=True):
        """tf.contrib.framework.assign_from_checkpoint_fn(trainable vars, ignore_missing)."""
        cur = self._named(self.engine.theta)
        for k, v in state['variables'].items():
            if k in cur and np.shape(cur[k]) == np.shape(v):
                cur[k] = v
            elif not ignore_missing:
                raise KeyError(k)
        self.engine.set_theta(self._pack(cur))
