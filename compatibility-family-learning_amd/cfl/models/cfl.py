"""The CFL model, distance phase (cfl/models/cfl.py:364-949, 1065-1085, 1348-1482)
for ``--model-type linear``: FCPCD encoders (weight-normalised heads,
cfl/models/blocks.py:477-527 + cfl/models/base.py:43-105), pcd / monomer / siamese
distance, learned-threshold BCE with pos_weight, lambda_m pull term or the
caffe-margin contrastive hinge, L2 regulariser, Adam (+ the separate threshold
Adam when --use-threshold is off) -- all inside the fused HIP pair kernels.

``--model-type conv`` adds the ConvPCD trunk (cfl.models.conv_encoder, weight-normalised
5x5 stride-2 convolutions on the GPU) in front of the same heads.

``--gan`` adds the MrCGAN post epochs (cfl/models/cfl.py:730-806, 951-1063, 1087-1096,
1484-1504): generator / discriminator stacks and the post-epoch step of cfl.models.mrcgan,
conditioned on the frozen distance encoder (cfl.models.encoder_heads).  Image and
image+latent ("double") datasets feed it; ``--cgan`` selects the conditional-GAN baseline (optionally ``--t-dim``).
"""
import logging
import os

import numpy as np

from .. import hipabi as H
from ..utils import dist_eval, load_best_stats, reduce_product, save_best_stats
from .base import PairModel

logger = logging.getLogger(__name__)


class CFL(PairModel):
    MODEL_SCOPE = 'CFL'
    HEAD_SCOPES = {
        'outputs': ('outputs/fully_connected', {'W': 'V', 'b': 'biases', 'g': 'g'}),
        'proto': ('prototype_outputs/fully_connected', {'W': 'V', 'b': 'biases', 'g': 'g'}),
        'mono': ('monomer_outputs/fully_connected', {'W': 'V', 'g': 'g'}),
    }

    def __init__(self, is_double, disable_double, latent_shape, source_shape, input_shape, ae_shape,
                 batch_size, data_norm, data_type, num_components, pos_weight, latent_size,
                 caffe_margin, gan, cgan, t_dim, dist_type, act_type, use_threshold, lr, beta1,
                 beta2, z_dim, z_stddev, g_dim, g_lr, g_beta1, g_beta2, m_prj, m_enc, d_dim, d_lr,
                 d_beta1, d_beta2, lambda_gp, lambda_m, lambda_dra, directed, data_directed,
                 model_type, gan_type, reg_const, batches=None, val_batches=None,
                 unlabeled_batches=None, train_data_transformer=None, val_data_transformer=None,
                 ae_transformer=None, data_normalizer=None, data_unnormalizer=None,
                 ae_normalizer=None, ae_unnormalizer=None, latent_normalizer=None, run_tag=None,
                 name='CFL', reuse=False, seed=0, device=None, layer_sizes=None):
        # layer_sizes: the hidden weight-normalised fc_i + lrelu layers of FCPCD (cfl/models/blocks.py:509-527; linear model only).
        # No command line of the reference sets it (its CFL class builds FCPCD without it, cfl/models/cfl.py:576-612); kept as a
        # constructor argument for callers of the model classes.
        for k, v in list(locals().items()):
            if k not in ('self', 'name', 'reuse', 'batches', 'val_batches', 'unlabeled_batches'):
                setattr(self, k, v)
        self.input_shape = tuple(input_shape)
        self.ae_shape = tuple(ae_shape) if ae_shape else self.input_shape
        self.source_shape = tuple(source_shape) if source_shape else self.input_shape
        if model_type not in ('linear', 'conv'):
            raise ValueError(model_type)
        self.ENCODER_SCOPES = ('DistEncoderSrc', 'DistEncoderDst') if directed else ('DistEncoder',)
        # double data: the encoder reads the pre-computed latents unless --data-disable-double
        # (cfl/models/cfl.py:176-193, 578)
        self.uses_latent = bool(is_double and not disable_double)
        if self.uses_latent:
            if not latent_shape:
                raise ValueError('double data needs --latent-shape')
            self.latent_shape = tuple(latent_shape) if not isinstance(latent_shape, int) else (latent_shape,)
            enc_norm = latent_normalizer
        else:
            enc_norm = data_normalizer
        # a per-channel normaliser cannot ride in the pair kernels' scalar affine map: it is applied as an explicit
        # pass on the rows (after the transformer) and the kernels see the identity
        self._explicit_norm = enc_norm if (enc_norm is not None and enc_norm.per_channel) else None
        norm = H.make_norm() if (enc_norm is None or self._explicit_norm is not None) else enc_norm.to_cfl_norm()
        loss = H.make_loss(use_threshold=use_threshold, pos_weight=pos_weight,
                           caffe_margin=caffe_margin, lambda_m=lambda_m, reg_const=reg_const)
        self.trunk = self.trunk_dst = None
        head_inputs = reduce_product(self.latent_shape if self.uses_latent else self.input_shape)
        if model_type == 'conv':
            # ConvPCD: the normaliser applies to the pixels; the heads see the flattened trunk
            from .conv_encoder import ConvTrunk
            import torch
            dev = torch.device(device if device is not None else 'cuda')
            shape3 = self.input_shape if len(self.input_shape) == 3 else self.input_shape + (1,)
            self.trunk = ConvTrunk(shape3, 4 * batch_size, enc_norm, reg_const, lr, beta1, beta2, 1e-8,
                                   np.random.RandomState(seed + 1), dev)
            # --directed: a second trunk for the target encoder (DistEncoderDst, cfl/models/cfl.py:676-681)
            self.trunk_dst = ConvTrunk(shape3, 4 * batch_size, enc_norm, reg_const, lr, beta1, beta2, 1e-8,
                                       np.random.RandomState(seed + 5), dev) if directed else self.trunk
            self._explicit_norm = None     # the trunk applies the data normaliser itself
            head_inputs, norm = self.trunk.feature_size, H.make_norm()
        elif layer_sizes:
            # FCPCD(layer_sizes=...): hidden fc_i layers in front of the heads, the same trunk plumbing as the conv encoder
            from .conv_encoder import FCTrunk
            import torch
            dev = torch.device(device if device is not None else 'cuda')
            mk = lambda sd: FCTrunk(head_inputs, layer_sizes, 4 * batch_size, enc_norm, reg_const, lr, beta1, beta2, 1e-8,
                                    np.random.RandomState(sd), dev)
            self.trunk = mk(seed + 1)
            self.trunk_dst = mk(seed + 5) if directed else self.trunk
            self._explicit_norm = None
            head_inputs, norm = self.trunk.feature_size, H.make_norm()
        self._setup_engine(
            head_inputs, latent_size, num_components, dist_type,
            weight_norm=True, has_bias=dist_type.startswith('pcd'), act_type=act_type,
            directed=directed, norm=norm, loss=loss, lr=lr, beta1=beta1, beta2=beta2,
            batch_size=batch_size, seed=seed, device=device)
        self._ema = {}
        self._np_rng = np.random.RandomState(int(seed) + 4)    # random crops / mirrors
        self._np_rng_shard = None                               # ... of a rank's rows of a host batch (data parallelism: train())
        self.seed = seed
        self.gan_phase = None
        self._gan_shard = None
        if gan:
            from .mrcgan import GanPhase
            from .. import engine as E
            # Data-parallel post epochs: every rank builds the inputs of the GLOBAL batch (encoder heads on B rows:
            # microseconds) and runs the G / D step on its rows -- `batch_size / world` of them -- then ONE all-reduce of
            # [d gradient | g gradient | scalars] in front of the two Adams.  Needs equal shares (--cgan: of each HALF of the batch,
            # its interpolation pairs row i with row i + B/2); otherwise every rank runs the whole batch (replicas, as round 5).
            # CFL_GAN_DP_SHARD=0: replicas.
            gan_rows = batch_size
            if E.dp_active() and os.environ.get('CFL_GAN_DP_SHARD', '1') not in ('0', ''):
                w = E.world_size()
                if batch_size % ((2 if cgan else 1) * w) == 0:
                    self._gan_shard, gan_rows = (E.rank(), w), batch_size // w
                else:
                    logger.warning('post epochs: batch size %d is not a multiple of %d: every rank runs the whole batch',
                                   batch_size, (2 if cgan else 1) * w)
            self.gan_phase = GanPhase(
                gan_type, self.ae_shape if len(self.ae_shape) == 3 else self.ae_shape + (1,), data_type, z_dim,
                latent_size, gan_rows, self.device, np.random.RandomState(seed + 2), g_lr=g_lr, g_beta1=g_beta1,
                g_beta2=g_beta2, d_lr=d_lr, d_beta1=d_beta1, d_beta2=d_beta2, lambda_gp=lambda_gp,
                lambda_dra=lambda_dra, m_enc=m_enc, m_prj=m_prj, cgan=cgan,
                c_dim=((reduce_product(self.input_shape) if self.trunk is not None else head_inputs) if t_dim
                       else latent_size) if cgan else None, t_dim=t_dim if cgan else None)
            if self._gan_shard is not None:
                self.gan_phase.shard_over(E.reduce_gradients)
            self._heads = None
            import torch
            self._gen = torch.Generator(device=self.device)
            self._gen.manual_seed(int(seed) + 3)

    def init(self, sess=None):
        pass

    # -- double data ---------------------------------------------------------------
    def select_batch(self, batch):
        """The 4 encoder inputs of a labeled batch: 8-tuples (image, latent interleaved,
        cfl/input_data.py:581-584) reduce to the latents, or to the images with --data-disable-double."""
        if len(batch) == 8:
            o = 1 if self.uses_latent else 0
            return (batch[o], batch[2 + o], batch[4 + o], batch[6 + o])
        return batch

    def select_pair(self, b):
        """(src, dst) encoder inputs of a whole_pos/neg_batches chunk (cfl/utils.py:234-244)."""
        if self.is_double:
            o = 1 if self.uses_latent else 0
            return b[o], b[2 + o]
        return b[0], b[1]

    # -- MrCGAN post-epoch iteration (cfl/models/cfl.py:1487-1497) -----------------------
    def _dev(self, x):
        import torch
        if not isinstance(x, torch.Tensor):
            return self.upload(x)
        return x.to(self.device, torch.float32).contiguous()

    def _enc_rows(self, x, side=0):
        """What the encoder heads read for raw input rows x: the padded rows themselves (linear model) or the
        flattened ConvPCD trunk features of the normalised pixels (conv model; frozen in the post epochs).
        side 1 = the target encoder's trunk when --directed."""
        if self.trunk is None:
            return self.to_device(self._prep(x, True))
        tr = self.trunk_dst if side else self.trunk
        return tr.forward(self._pixels(x, True)).clone()   # a trunk reuses its activation buffers per row count

    def _draws(self, draws):
        """(z, eps, c) of one post-epoch iteration: z ~ N(0, z_stddev), eps ~ U[0,1), gate c ~ U{0..K-1}
        (cfl/models/cfl.py:70-100, 535-546), or the caller's values (feeding the reference's z / eps / c
        placeholders_with_default)."""
        import torch
        B = self.batch_size
        if draws is not None:
            z, eps, c = draws
            return (self._dev(np.asarray(z, np.float32)), self._dev(np.asarray(eps, np.float32).reshape(B, 1)),
                    torch.as_tensor(np.asarray(c), dtype=torch.int32, device=self.device))
        c = torch.randint(0, self.num_components, (B,), generator=self._gen, device=self.device, dtype=torch.int32)
        z = torch.randn(B, self.z_dim, generator=self._gen, device=self.device) * float(self.z_stddev)
        eps = torch.rand(B, 1, generator=self._gen, device=self.device)
        return z.contiguous(), eps, c

    def gan_inputs(self, labeled, unl_src, unl_dst, draws=None):
        """Device inputs of GanPhase.step from one labeled batch and the unlabeled source / target item
        batches (lists [x] or [image, latent])."""
        import torch
        from .. import hipgan as G
        from .encoder_heads import FrozenHeads
        if self._heads is None:
            self._heads = FrozenHeads(self.engine, self.act_type)
        hd, B = self._heads, self.batch_size
        o = 1 if self.uses_latent else 0
        enc_in = lambda parts, side=0: self._enc_rows(parts[o] if len(parts) > 1 else parts[0], side)
        lab = self.select_batch(labeled)
        dst_side = 1 if self.directed else 0
        real = self._ae_image(unl_dst[0])
        z, eps, c = self._draws(draws)
        enc_act = hd.activations(enc_in(unl_dst, dst_side), dst_side)
        prj_c = G.gather_prototype(hd.prototype_activations(enc_in(unl_src), 0), c)
        neg_c = G.gather_prototype(hd.prototype_activations(self._enc_rows(lab[2]), 0), c)
        neg_tgt_act = hd.activations(self._enc_rows(lab[3], dst_side), dst_side)
        return real, enc_act, prj_c, neg_c, neg_tgt_act, z, eps

    def cgan_inputs(self, labeled, draws=None):
        """Device inputs of GanPhase.step_cgan (cfl/models/cfl.py:747-782): positive / negative TARGET images
        and the source-side conditions -- the source encoder's activations, or with --t-dim the (normalised)
        source inputs themselves."""
        import torch
        from .. import hipgan as G
        from .encoder_heads import FrozenHeads
        if self._heads is None:
            self._heads = FrozenHeads(self.engine, self.act_type)
        hd, B = self._heads, self.batch_size
        lab = self.select_batch(labeled)
        per = 2 if len(labeled) == 8 else 1
        real_pos, real_neg = self._ae_image(labeled[1 * per]), self._ae_image(labeled[3 * per])
        cond = self._cgan_condition
        z, eps, _ = self._draws(draws)
        return real_pos, real_neg, cond(lab[0]), cond(lab[2]), z, eps

    def _cgan_condition(self, a):
        """pos_src / neg_src of cfl/models/cfl.py:748-749: the source encoder's activations, or with --t-dim the
        normalised encoder input itself (pixels for the conv model)."""
        hd = self._sample_heads()
        if not self.t_dim:
            return hd.activations(self._enc_rows(a), 0)
        if self.trunk is not None:
            return self._pixels(a, True)
        return hd.normalize(self.to_device(self._prep(a, True)))[:, :self.input_size].contiguous()

    def _gan_rows(self, ins):
        """this rank's rows of the global batch's GanPhase inputs (+ X_hat, whose std is a statistic of the GLOBAL batch)"""
        import torch
        from .. import hipgan as G
        gp, (r, w), B = self.gan_phase, self._gan_shard, self.batch_size
        x_hat = G.perturb(ins[0], ins[-1], gp.lambda_dra) if gp.lambda_gp else None
        if self.cgan:       # rows [r hs, (r+1) hs) of BOTH halves: the interpolated conditions pair row i with row i + B/2
            hs = B // 2 // w
            if getattr(self, '_gan_idx', None) is None:
                a = torch.arange(r * hs, (r + 1) * hs, device=self.device)
                self._gan_idx = torch.cat([a, a + B // 2])
            take = lambda t: t.index_select(0, self._gan_idx)
        else:
            lo, hi = r * (B // w), (r + 1) * (B // w)
            take = lambda t: t[lo:hi].contiguous()
        return tuple(take(t) for t in ins), (take(x_hat) if x_hat is not None else None)

    def post_step(self, labeled, unl_src=None, unl_dst=None, draws=None):
        ins = self.cgan_inputs(labeled, draws) if self.cgan else self.gan_inputs(labeled, unl_src, unl_dst, draws)
        x_hat = None
        if self._gan_shard is not None:
            ins, x_hat = self._gan_rows(ins)
        if self.cgan:
            self.gan_phase.step_cgan(*ins, x_hat=x_hat)
        else:
            self.gan_phase.step(*ins, x_hat=x_hat)

    # -- sampling (cfl/models/cfl.py:808-860: s_encoder_sample, g_target, g_prototypes, d_prototypes) --------
    def _sample_heads(self):
        from .encoder_heads import FrozenHeads
        if self._heads is None:
            self._heads = FrozenHeads(self.engine, self.act_type)
        return self._heads

    def _sample_z(self, n):
        import torch
        return (torch.randn(n, self.z_dim, generator=self._gen, device=self.device) * float(self.z_stddev)).contiguous()

    def generate_prototypes(self, src_rows):
        """[G(z, prototype_k(src)) for k < K] and the discriminator's sigmoid outputs on them; `src_rows` are
        raw encoder inputs [n, D] (n = batch_size).  One z is shared by all components, as in the reference
        graph (self.z)."""
        import torch
        from .. import hipgan as G
        hd = self._sample_heads()
        images, preds = [], []
        if self.cgan:
            # cgan: one condition, a fresh z per "prototype" (self.zs[i], cfl/models/cfl.py:831-846)
            c = self._cgan_condition(src_rows)
            for k in range(self.num_components):
                x = c
                acts = self.gan_phase.generate(self._sample_z(x.shape[0]), c)
                logit, _, _ = self.gan_phase.disc.forward(acts, c)
                images.append(acts.cpu().numpy())
                preds.append(G.act_fwd(logit.contiguous(), 'sigmoid').cpu().numpy())
            return images, preds
        x = self._enc_rows(src_rows)
        P = hd.prototype_activations(x, 0)
        z = self._sample_z(x.shape[0])
        for k in range(self.num_components):
            c = torch.full((x.shape[0],), k, dtype=torch.int32, device=self.device)
            acts = self.gan_phase.generate(z, G.gather_prototype(P, c))
            logit, _, _ = self.gan_phase.disc.forward(acts)
            images.append(acts.cpu().numpy())
            preds.append(G.act_fwd(logit.contiguous(), 'sigmoid').cpu().numpy())
        return images, preds

    def generate_target(self, dst_rows):
        """G(z, activations of the target encoder) (self.g_target)."""
        hd = self._sample_heads()
        x = self._enc_rows(dst_rows, 1 if self.directed else 0)
        acts = self.gan_phase.generate(self._sample_z(x.shape[0]), hd.activations(x, 1 if self.directed else 0))
        return acts.cpu().numpy()

    # -- input transformers (cfl/models/cfl.py:136-147, 309-320; cfl/ops.py:262-299) ------------------------
    def _prep(self, x, train):
        """Raw dataset rows -> the rows the encoder / normaliser sees: the train (random crop / mirror) or val
        (central crop / resize) transformer on image-shaped inputs; latents and untransformed data pass through."""
        tr = self.train_data_transformer if train else self.val_data_transformer
        if self.uses_latent or (tr is None and self._explicit_norm is None):
            return x
        t = self._dev(x)
        if tr is not None and t.shape[1] == int(np.prod(tr.source_shape)):
            t = tr.apply(t, self._np_rng)
        if self._explicit_norm is not None:
            t = self._explicit_norm.apply(t)
        return t

    def _ae_image(self, x):
        """ae-normalised image rows for the generator / discriminator: transformer, optional resize to ae_shape
        (dist_ae_transformer), ae_normalizer."""
        from .. import hipgan as G
        tr = self.train_data_transformer
        t = self._dev(x)
        if tr is not None and t.shape[1] == int(np.prod(tr.source_shape)):
            t = tr.apply(t, self._np_rng)
        if self.ae_transformer is not None:
            t = self.ae_transformer.apply(t, self._np_rng)
        return self.ae_normalizer.apply(t) if self.ae_normalizer is not None else t

    # -- ConvPCD: trunk + heads ---------------------------------------------------
    def _pixels(self, x, train=False):
        import torch
        x = self._prep(x, train)
        t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, np.float32))
        return self.trunk.normalize(t.to(self.device, torch.float32))

    def train_step(self, batch):
        if self.is_indexed(batch):
            return PairModel.train_step(self, batch)
        if self._np_rng_shard is None:
            return self._train_step_host(batch)
        shared, self._np_rng = self._np_rng, self._np_rng_shard
        try:
            return self._train_step_host(batch)
        finally:
            self._np_rng = shared

    def _train_step_host(self, batch):
        batch = self.select_batch(batch)
        if self.trunk is None:
            return PairModel.train_step(self, [self._prep(b, True) for b in batch])
        import torch
        eng = self.engine
        B = batch[0].shape[0]
        # rows ordered [pos_src, neg_src, pos_dst, neg_dst]: each side's 2B rows contiguous
        xs = torch.cat([self._pixels(batch[0], True), self._pixels(batch[2], True)])
        xd = torch.cat([self._pixels(batch[1], True), self._pixels(batch[3], True)])
        two = self.trunk_dst is not self.trunk
        if two:
            Fs, Fd = self.trunk.forward(xs), self.trunk_dst.forward(xd)
        else:
            F = self.trunk.forward(torch.cat([xs, xd]))
            Fs, Fd = F[0:2 * B], F[2 * B:4 * B]
        eng.fwd_bwd((Fs[0:B], Fd[0:B], Fs[B:2 * B], Fd[B:2 * B]))
        ws = eng._workspace(B, 2)
        if two:
            dFs, dFd = torch.empty_like(Fs), torch.empty_like(Fd)
            H.pair_input_grad(eng.shape, eng.norm, B, eng.theta, ws, dFs, dFd)
            self.trunk.backward(dFs)
            self.trunk_dst.backward(dFd)
        else:
            dF = torch.empty_like(F)
            H.pair_input_grad(eng.shape, eng.norm, B, eng.theta, ws, dF[0:2 * B], dF[2 * B:4 * B])
            self.trunk.backward(dF)
        lr_t = eng.lr_t()
        scale = 1.0
        trunks = [self.trunk, self.trunk_dst] if two else [self.trunk]
        if eng.world_size > 1:
            from ..engine import reduce_gradients
            scale = eng._scalar_scale = reduce_gradients(eng.gradbuf)     # gradient + the step's scalars
            for tr in trunks:
                reduce_gradients(tr.grad)
        for tr in trunks:
            tr.apply_adam(lr_t, scale)
        eng.apply_adam(scale)

    def predict(self, src, dst):
        if self.trunk is None:
            return PairModel.predict(self, self._prep(src, False), self._prep(dst, False))
        import torch
        n = src.shape[0]
        if self.trunk_dst is not self.trunk:
            Fs, Fd = self.trunk.forward(self._pixels(src)), self.trunk_dst.forward(self._pixels(dst))
            return self.engine.scores(Fs, Fd).cpu().numpy().reshape(-1, 1)
        F = self.trunk.forward(torch.cat([self._pixels(src), self._pixels(dst)]))
        return self.engine.scores(F[:n], F[n:]).cpu().numpy().reshape(-1, 1)

    def batch_accuracy(self, batch):
        if self.is_indexed(batch):
            return PairModel.batch_accuracy(self, batch)
        batch = self.select_batch(batch)
        if self.trunk is None:
            return PairModel.batch_accuracy(self, [self._prep(b, False) for b in batch])
        sp, sn = self.predict(batch[0], batch[1]), self.predict(batch[2], batch[3])
        return 0.5 * float((sp > 0).mean() + (sn <= 0).mean())

    def assign_trainable(self, state, ignore_missing=True):
        """--load-pre-weights (cfl/utils.py:480-494): encoder variables of the no-gan run, incl. the conv trunks."""
        PairModel.assign_trainable(self, state, ignore_missing)
        for i, tr in enumerate(self._trunks()):
            pre = 'CFL/' + self.ENCODER_SCOPES[i] + '/'
            tr.load_named({k[len(pre):]: v for k, v in state['variables'].items() if k.startswith(pre + tr.PREFIX)})

    def _trunks(self):
        if self.trunk is None:
            return []
        return [self.trunk, self.trunk_dst] if self.trunk_dst is not self.trunk else [self.trunk]

    def scalars(self):
        s = PairModel.scalars(self)
        if self.trunk is not None and self.reg_const:
            extra = self.trunk.reg_loss()     # conv V regulariser (cfl/models/blocks.py:585)
            if self.trunk_dst is not self.trunk:
                extra += self.trunk_dst.reg_loss()
            s['reg'] += extra
            s['total'] += extra
        return s

    def checkpoint_state(self):
        st = PairModel.checkpoint_state(self)
        for i, tr in enumerate(self._trunks()):
            pre = 'CFL/' + self.ENCODER_SCOPES[i] + '/'
            for key, base in (('variables', None), ('adam_m', tr.m), ('adam_v', tr.v)):
                st[key].update({pre + k: v for k, v in tr.named(base).items()})
        if self.gan_phase is not None:
            for net in (self.gan_phase.gen, self.gan_phase.disc):
                ns = net.state()
                for key in ('variables', 'adam_m', 'adam_v'):
                    st[key].update({'CFL/' + k: v for k, v in ns[key].items()})
            st['gan_powers'] = {n: (float(net.beta1_power), float(net.beta2_power))
                                for n, net in (('g', self.gan_phase.gen), ('d', self.gan_phase.disc))}
        return st

    def load_checkpoint_state(self, state):
        PairModel.load_checkpoint_state(self, state)
        for i, tr in enumerate(self._trunks()):
            pre = 'CFL/' + self.ENCODER_SCOPES[i] + '/'
            for key, base in (('variables', None), ('adam_m', tr.m), ('adam_v', tr.v)):
                tr.load_named({k[len(pre):]: v for k, v in state[key].items() if k.startswith(pre + tr.PREFIX)}, base)
        if self.gan_phase is not None and 'gan_powers' in state:
            for n, net in (('g', self.gan_phase.gen), ('d', self.gan_phase.disc)):
                strip = lambda d: {k[4:]: v for k, v in d.items() if k.startswith('CFL/')}
                net.load_state({'variables': strip(state['variables']), 'adam_m': strip(state['adam_m']),
                                'adam_v': strip(state['adam_v']), 'beta1_power': state['gan_powers'][n][0],
                                'beta2_power': state['gan_powers'][n][1]})

    def get_name(self, no_gan=False):
        """Byte-for-byte cfl/models/cfl.py:368-412 (names checkpoint / predict dirs)."""
        parts = ['cfl', self.dist_type, self.model_type]
        if self.directed:
            parts.append('di')
        if self.pos_weight:
            parts.append('pw_{}'.format(self.pos_weight))
        if self.caffe_margin:
            parts.append('margin_{}'.format(self.caffe_margin))
        parts += [self.data_type, 'ls_{}'.format(self.latent_size)]
        if self.dist_type != 'siamese':
            parts.append('nc_{}'.format(self.num_components))
        if self.act_type:
            parts.append('act_{}'.format(self.act_type))
        if self.disable_double:
            parts.append('dd')
        if self.use_threshold:
            parts.append('ut')
        if self.reg_const:
            parts.append('reg_{}'.format(self.reg_const))
        if self.data_norm:
            parts.append('norm_{}'.format('_'.join(str(n) for n in self.data_norm)))
        if self.lambda_m:
            parts.append('lm_{}'.format(self.lambda_m))
        if self.gan and not no_gan:
            if self.cgan:
                parts.append('cgan_z_{}'.format(self.z_dim))
                if self.t_dim:
                    parts.append('t_{}'.format(self.t_dim))
            else:
                parts.append('gan_z_{}'.format(self.z_dim))
                if self.m_prj:
                    parts.append('m_prj_{}'.format(self.m_prj))
                if self.m_enc:
                    parts.append('m_enc_{}'.format(self.m_enc))
            if self.lambda_gp:
                parts.append('dra_{}_{}'.format(self.lambda_gp, self.lambda_dra))
            if self.gan_type != 'conv':
                parts.append(self.gan_type)
        name = '_'.join(parts)
        if self.run_tag:
            name += '_run_' + self.run_tag
        return name

    # ExponentialMovingAverage(0.99), zero-initialised shadow, no debias
    # (cfl/models/cfl.py:528, 896-949; display only)
    def _ema_update(self, key, value):
        self._ema[key] = 0.99 * self._ema.get(key, 0.0) + 0.01 * value
        return self._ema[key]

    def train(self, sess, data, start_iter, epochs, post_epochs, best_dir, best_acc_dir,
              checkpoint_dir, epoch_callback=None, post_epoch_callback=None, save_epochs=1,
              eval_epochs=1, save_iters=None, disable_eval=False, saver=None, best_saver=None,
              best_acc_saver=None, writer=None, check=None, val_every=1):
        """Distance epochs of cfl/models/cfl.py:1348-1511 (same bookkeeping, same
        best_model / best_acc_model files, including the reference's habit of
        writing the AUC-best stats into best_accuracy_by_th)."""
        from tqdm import trange
        nb_train = max(data.train.num_examples_labeled_pos, data.train.num_examples_labeled_neg)
        logger.warning('%d pairs / %d images', nb_train, data.train.num_examples)
        # labeled batches of an image + latent dataset: this model reads their latents only (select_batch; the post epochs take
        # their images from the unlabeled streams) unless it is the --cgan baseline, which trains on the labeled images
        if self.is_double and self.uses_latent and not self.cgan:
            for split in (data.train, data.val):
                if hasattr(split, 'labeled_images'):
                    split.labeled_images = False
        nb_batch = nb_train // self.batch_size
        logger.warning('%d batches per epoch', nb_batch)
        best_auc_path = os.path.join(best_dir, 'best_accuracy')
        best_acc_path = os.path.join(best_acc_dir, 'best_accuracy_by_th')
        stats = load_best_stats(best_auc_path)
        stats_acc = load_best_stats(best_acc_path)
        if not self.gan:
            logger.info('post epochs disabled due to disabled gan')
            post_epochs = 0
        total_epochs = epochs + post_epochs
        start_epoch = start_iter // nb_batch
        logger.warning('start epoch %d of %d', start_epoch, total_epochs)
        # vector datasets: features.b of a split lives in HBM and batches are gathered there by index
        # (cfl_gather_rows); under torchrun every rank walks the same seeded index stream and trains on
        # its own slice of each global batch
        from .. import engine as dp
        from ..input_data import feature_source
        resident = None
        if not data.train.is_image and self.trunk is None and self.train_data_transformer is None \
                and self.val_data_transformer is None and self._explicit_norm is None:
            resident = (feature_source(data.train, self.device), feature_source(data.val, self.device))
        elif data.train.is_image and data.train.is_double and self.uses_latent and self.trunk is None \
                and os.environ.get('CFL_DOUBLE_RESIDENT', '1') not in ('0', ''):
            # image + latent dataset, encoder on the latents: the latents of every record live in HBM and the distance epochs
            # run on the indexed kernels as for a vector dataset (the same dataset object keeps driving the seeded streams, so
            # the post epochs continue the reference's draw sequence)
            resident = (feature_source(data.train, self.device), feature_source(data.val, self.device))
        shard = dp.shard_rows(self.batch_size) if dp.world_size() > 1 else None
        chief = dp.rank() == 0
        if shard is not None and resident is None and self._np_rng_shard is None:
            # host batches (image data, conv trunk, per-batch transformers) under data parallelism: every rank draws the SAME
            # global batch from the seeded streams and trains on its rows of it; the random crops / mirrors of those rows come
            # from a stream of the rank's own (the ranks' rows must not share their crop offsets)
            self._np_rng_shard = np.random.RandomState((int(self.seed) + 4 + 7919 * (dp.rank() + 1)) % (2 ** 31))

        def next_train():
            if resident:
                return resident[0].next_indexed(self.batch_size, shard)
            b = data.train.next_batch(self.batch_size)
            return b if shard is None else [x[shard[0]:shard[1]] for x in b]

        def next_val():
            return resident[1].next_indexed(self.batch_size) if resident else data.val.next_batch(self.batch_size)
        for e in range(start_epoch, total_epochs):
            t = trange(start_iter % nb_batch if e == start_epoch else 0, nb_batch, disable=not chief)
            if e >= epochs:
                # every rank, before the chief-only saves of the post epochs (the encoder's Adam slots are sharded under
                # the one-shot exchange; a no-op when nothing was stepped since the last sync)
                self.engine.sync_state()
                self._post_epoch(e, t, data, nb_batch, save_iters, saver if chief else None, checkpoint_dir, writer)
                if e % save_epochs == 0 and saver is not None and chief:
                    saver.save(self, os.path.join(checkpoint_dir, 'model'), global_step=(e + 1) * nb_batch)
                continue
            t.set_description('epoch {}'.format(e))
            train_avg = val_avg = 0.0
            # Resident features + the linear encoder: a whole epoch goes through cfl.bin.train_dist.train_steps -- windows of the
            # device pair lists through the fused multi-iteration library call, read-backs that do not stall the stream (the same
            # iterations, 0, 50, 100, ... and the last, with the validation batch inside the step's launches where the library
            # can) -- instead of one Python iteration per step (measured on the dyadic-generation shape: 0.47 ms of host time per
            # 45 us step).  Same draws from the same seeded streams, same trajectory (step windows == single indexed steps bit for
            # bit).  Not with --save-iters or when resuming inside an epoch (the cadence is counted from the epoch's first
            # iteration); CFL_FUSED_EPOCHS=0: the per-iteration loop.
            if (resident is not None and self.trunk is None and not save_iters and hasattr(self.engine, 'step_windows')
                    and (start_iter % nb_batch == 0 or e != start_epoch)
                    and os.environ.get('CFL_FUSED_EPOCHS', '1') not in ('0', '')):
                from ..bin.train_dist import train_steps
                bad, avgs = [], {}

                def on_scalars(i, s, val_acc, e=e, t=t):
                    if not np.isfinite(s['total']):
                        bad.append(nb_batch * e + i)
                    if writer is not None and chief:
                        writer.add_scalars('scalars', nb_batch * e + i, s)
                    avgs['train'] = self._ema_update('acc', s['accuracy'])
                    avgs['val'] = self._ema_update('val_acc', val_acc)
                    t.set_postfix(error=1. - avgs['train'], val_error=1. - avgs['val'],
                                  pos_avg=self._ema_update('pos', s['dist_adapt_pos']),
                                  neg_avg=self._ema_update('neg', s['dist_adapt_neg']))
                train_steps(self, resident[0], resident[1], self.batch_size, shard, nb_batch, on_scalars, progress=t,
                            scalar_every=50)
                t.close()
                if bad:
                    # never checkpoint poisoned parameters: the last files written stay the latest ones
                    raise FloatingPointError('non-finite training loss at iteration {}'.format(bad[0]))
                train_avg, val_avg = avgs.get('train', 0.0), avgs.get('val', 0.0)
                t = ()
            for i in t:
                rb = i % 50 == 0 or i == nb_batch - 1
                batch = next_train()
                # the validation batch of a read-back iteration is scored with the weights BEFORE that iteration's update -- what the
                # fused epochs above do, and what one sess.run of the reference fetches (cfl/models/cfl.py:1399-1414): ONE semantics
                val_acc = self.batch_accuracy(next_val()) if rb else None
                self.train_step(batch)
                if save_iters and i > 0 and i % save_iters == 0 and saver is not None:
                    self.engine.sync_state()      # collective when the Adam slots are sharded (one-shot exchange)
                    if chief:
                        saver.save(self, os.path.join(checkpoint_dir, 'model'), global_step=nb_batch * e + i)
                if rb:                                    # host read-back only now and then
                    s = self.scalars()                    # (raises CflHipError on a lost in-launch hand-off)
                    if not np.isfinite(s['total']):
                        # never checkpoint poisoned parameters: the last files written stay the latest ones
                        raise FloatingPointError('non-finite training loss at iteration {}'.format(nb_batch * e + i))
                    if writer is not None and chief:
                        writer.add_scalars('scalars', nb_batch * e + i, s)
                    train_avg = self._ema_update('acc', s['accuracy'])
                    val_avg = self._ema_update('val_acc', val_acc)
                    t.set_postfix(error=1. - train_avg, val_error=1. - val_avg,
                                  pos_avg=self._ema_update('pos', s['dist_adapt_pos']),
                                  neg_avg=self._ema_update('neg', s['dist_adapt_neg']))
            # evaluation is collective under data parallelism (utils._sharded_eval): every rank evaluates and
            # keeps the same best-model bookkeeping, rank 0 alone writes
            if e % eval_epochs == 0 and not disable_eval:
                val_stats = dist_eval(None, self, self.batch_size, data.val)
                if val_stats.auc > stats.best_auc or val_stats.accuracy > stats_acc.best_accuracy:
                    test_stats = dist_eval(None, self, self.batch_size, data.test)
                    logger.warning('epoch %d: current error = train: %f val: %f test: %f / auc = val: %f test: %f',
                                   e, 1. - train_avg, 1. - val_stats.accuracy, 1. - test_stats.accuracy,
                                   val_stats.auc, test_stats.auc)
                    if val_stats.auc > stats.best_auc:
                        stats.best_accuracy, stats.best_auc, stats.best_epoch = \
                            val_stats.accuracy, val_stats.auc, e
                        self.engine.sync_state()      # collective when the Adam slots are sharded (one-shot exchange)
                        if chief:
                            best_saver.save(self, os.path.join(best_dir, 'model'), global_step=stats.best_epoch)
                            save_best_stats(best_auc_path, stats.best_epoch, stats.best_accuracy, stats.best_auc)
                    if val_stats.accuracy > stats_acc.best_accuracy:
                        stats_acc.best_accuracy, stats_acc.best_auc, stats_acc.best_epoch = \
                            val_stats.accuracy, val_stats.auc, e
                        # reference quirk (cfl/models/cfl.py:1463-1470): step and file
                        # carry the AUC-best `stats`, not `stats_acc`
                        step = stats.best_epoch if stats.best_epoch is not None else e
                        self.engine.sync_state()      # collective when the Adam slots are sharded (one-shot exchange)
                        if chief:
                            best_acc_saver.save(self, os.path.join(best_acc_dir, 'model'), global_step=step)
                            save_best_stats(best_acc_path, stats.best_epoch, stats.best_accuracy, stats.best_auc)
                else:
                    logger.warning('epoch %d: current error = train: %f val: %f / auc = val: %f',
                                   e, 1. - train_avg, 1. - val_stats.accuracy, val_stats.auc)
            else:
                logger.warning('epoch %d: avg error = train: %f val: %f', e, 1. - train_avg, 1. - val_avg)
            if e % save_epochs == 0 and saver is not None:
                self.engine.sync_state()      # collective when the Adam slots are sharded (one-shot exchange)
                if chief:
                    saver.save(self, os.path.join(checkpoint_dir, 'model'), global_step=(e + 1) * nb_batch)


    def _post_epoch(self, e, t, data, nb_batch, save_iters, saver, checkpoint_dir, writer=None):
        """One MrCGAN post epoch (cfl/models/cfl.py:1484-1504): every iteration draws a labeled batch and
        an unlabeled source / target batch (cfl/bin/train.py:29-40) and updates D and G."""
        t.set_description('post epoch {}'.format(e))
        tr = data.train

        def fetch():
            """the batches of one iteration, in the reference's order of draws (cfl/bin/train.py:29-40)"""
            labeled = tr.next_batch(self.batch_size)
            if self.cgan:
                return (labeled,)
            if self.directed or self.data_directed:
                return labeled, tr.next_source_batch(self.batch_size), tr.next_target_batch(self.batch_size)
            unl = tr.next_unlabeled_batch(self.batch_size)
            return labeled, unl, unl
        # (tried: the batches of iteration i + 1 assembled by a worker thread while this thread enqueues iteration i's ~500
        # launches -- 15.3 -> 16.2 ms per iteration: both sides are interpreter-bound and take turns on its lock; not kept)
        for i in t:
            batches = fetch()
            self.post_step(*batches)
            if save_iters and i > 0 and i % save_iters == 0 and saver is not None:
                saver.save(self, os.path.join(checkpoint_dir, 'model'), global_step=nb_batch * e + i)
            if i % 20 == 0 or i == nb_batch - 1:
                s = self.gan_phase.read_scalars()
                if writer is not None:
                    writer.add_scalars('gan_scalars', nb_batch * e + i, s)
                t.set_postfix(d_loss=s['d_total_loss'], g_loss=s['g_total_loss'], d_real=s['d_real_accuracy'],
                              d_fake=s['d_fake_accuracy'])
        s = self.gan_phase.read_scalars()
        logger.warning('post epoch %d: d_total %f (real %f fake %f gp %f lat %f) g_total %f (adv %f enc %f neg %f)',
                       e, s['d_total_loss'], s['d_loss_real'], s['d_loss_fake'], s['d_grad_loss'], s['d_loss_d'],
                       s['g_total_loss'], s['g_loss'], s['g_loss_d'], s['g_loss_d_neg'])


def construct_model(is_double, disable_double, latent_shape, source_shape, input_shape, ae_shape,
                    batch_size, data_norm, data_type, model_type, gan_type, num_components,
                    latent_size, pos_weight, caffe_margin, gan, cgan, t_dim, dist_type, act_type,
                    use_threshold, lr, beta1, beta2, z_dim, z_stddev, g_dim, g_lr, g_beta1,
                    g_beta2, m_prj, m_enc, d_dim, d_lr, d_beta1, d_beta2, lambda_dra, lambda_gp,
                    lambda_m, directed, data_directed, reg_const, data=None, run_tag=None,
                    train_data_transformer=None, val_data_transformer=None, ae_transformer=None,
                    data_normalizer=None, data_unnormalizer=None, ae_normalizer=None,
                    ae_unnormalizer=None, latent_normalizer=None, enable_input_producer=False,
                    seed=0, device=None, layer_sizes=None):
    """(model, aux) with the argument list of cfl/models/cfl.py:1514-1523 (+ layer_sizes: FCPCD's hidden layers,
    cfl/models/blocks.py:492,509-524, which the reference's construct_model leaves at None)."""
    from argparse import Namespace
    kw = dict(locals())
    for k in ('data', 'enable_input_producer', 'Namespace'):
        kw.pop(k)
    model = CFL(**kw)
    aux = None if data is None else Namespace(train=data.train, unlabeled=data.train, val=data.val)
    return model, aux
