"""The MrCGAN post-epoch step on the GPU (MrCGAN branch and the conditional-GAN baseline `--cgan`).

Graph:   cfl/models/cfl.py:784-806   g = G(z, enc_dst(unlabeled target)), g_prj = G(z, one prototype of
                                      the unlabeled source), g_neg = G(z, one prototype of the negative
                                      source); D on real / g / g_prj / g_neg / X_hat
Losses:  cfl/models/cfl.py:951-1063
Update:  cfl/models/cfl.py:1087-1096 (two TF-Adams), run in ONE sess.run at cfl/models/cfl.py:1491-1497,
         i.e. both gradients are taken at the same (pre-update) weights.

The three generator evaluations share one batched pass (rows [g | g_prj | g_neg]) and the five
discriminator evaluations one batched forward (rows [real | g | g_prj | g_neg | X_hat]); weight
normalisation has no cross-sample statistics, so batching is exact.  Backward passes run on the row
ranges that carry gradient: d-loss on [real | g | g_prj] (variables only), g-loss on [g | g_prj | g_neg]
(inputs only, then through the generator), gradient penalty on [X_hat].

--cgan (cfl/models/cfl.py:747-782, 969-981, 1022-1038): g = G(z, c_pos), g_int = G(z[:B/2], (c_pos[:B/2] + c_pos[B/2:])/2);
D(x, t) on (pos target, c_pos) / (g, c_pos) / (neg target, c_neg) / (g_int, c_half) / (X_hat, c_pos);
d = BCE(real,1) + (BCE(fake,0) + BCE(neg,0))/2 + GP;  g = BCE(fake,1) + BCE(int,1).
"""
import os

import numpy as np
import torch

from .. import hipgan as G
from .gan_blocks import Discriminator, Generator

# indices into GanPhase.scalars (device floats); the cgan branch reuses S_D_PRJ for d_loss_neg and
# S_G_PRJ for g_loss_int
S_D_REAL, S_D_ENC, S_D_PRJ, S_D_GP, S_D_LAT, S_G_ENC, S_G_PRJ, S_G_LAT, S_G_NEG, S_FRAC_REAL, S_FRAC_FAKE, \
    S_FRAC_PRJ = range(12)


class GanPhase(object):
    gp_early = os.environ.get('CFL_GAN_GP_EARLY', '1') not in ('0', '')

    def __init__(self, gan_type, ae_shape, data_type, z_dim, latent_size, batch_size, device, rng,
                 g_lr=2e-4, g_beta1=0.5, g_beta2=0.999, d_lr=2e-4, d_beta1=0.5, d_beta2=0.999,
                 lambda_gp=None, lambda_dra=0.5, m_enc=None, m_prj=None, cgan=False, c_dim=None, t_dim=None):
        """cgan: conditional-GAN baseline; c_dim = width of the condition (latent_size, or the raw source
        latent size with t_dim, which then adds the fc_t layers)."""
        self.gan_type, self.ae_shape, self.data_type = gan_type, tuple(ae_shape), data_type
        self.z_dim, self.latent_size, self.B = z_dim, latent_size, batch_size
        self.device = device
        self.lambda_gp, self.lambda_dra, self.m_enc, self.m_prj = lambda_gp, lambda_dra, m_enc, m_prj
        self.cgan = bool(cgan)
        self.c_dim = (c_dim or latent_size) if cgan else latent_size
        if cgan:
            if batch_size % 2:
                raise ValueError('--cgan needs an even batch size (cfl/models/cfl.py:752)')
            self.gen = Generator(gan_type, ae_shape, z_dim + self.c_dim, data_type, rng, device, g_lr, g_beta1,
                                 g_beta2, c_dim=self.c_dim, t_dim=t_dim)
            self.disc = Discriminator(gan_type, ae_shape, latent_size, rng, device, d_lr, d_beta1, d_beta2,
                                      c_dim=self.c_dim, t_dim=t_dim)
        else:
            self.gen = Generator(gan_type, ae_shape, z_dim + latent_size, data_type, rng, device, g_lr, g_beta1,
                                 g_beta2)
            self.disc = Discriminator(gan_type, ae_shape, latent_size, rng, device, d_lr, d_beta1, d_beta2)
        # ONE flat buffer [discriminator gradient | generator gradient | 16 scalars (+ pad)]: under data parallelism the whole
        # exchange of a post-epoch iteration is one all-reduce of it (shard_over)
        nd, ng = self.disc.pool.total, self.gen.pool.total
        self._flat = torch.zeros(nd + ng + 64, dtype=torch.float32, device=device)
        self.disc.pool.grad, self.gen.pool.grad = self._flat[:nd], self._flat[nd:nd + ng]
        self.scalars = self._flat[nd + ng:nd + ng + 16]
        self.reduce = None      # data parallelism: flat buffer -> factor that turns the sums over the ranks into global-batch means
        self.ae_size = int(np.prod(self.ae_shape))
        # diagnostics: keep the (generator, discriminator) tapes of the last step alive (tests read the activation
        # signs from them); off by default so that a step's activations are released when it returns
        self.keep_tapes = False
        self.last_tapes = None
        self.timing = None      # set to {} to have step() record where its chains end (ms from the start of the step)

    def shard_over(self, reduce):
        """Data-parallel post epochs (rows are independent given the weights): this GanPhase was built with the
        rank's share B / world of the global batch, the caller hands every step the rank's ROWS of the global batch's inputs (and
        X_hat computed from the GLOBAL batch: its std is a statistic of all elements, cfl/models/cfl.py:742-745), and before the
        two Adams `reduce(flat)` sums [d gradient | g gradient | scalars] over the ranks in ONE collective and returns
        1 / world: every loss is a mean over equal shares, so the mean of the ranks' gradients is the global-batch gradient."""
        self.reduce = reduce

    def _exchange(self):
        """-> gradient scale of the two Adams (1.0 without data parallelism)"""
        if self.reduce is None:
            return 1.0
        scale = float(self.reduce(self._flat))
        self.scalars.mul_(scale)        # (16 floats: the logged losses / accuracies become global-batch means)
        return scale

    def generate(self, z, c):
        """Generator activations for display / sampling (cfl.bin.sample)."""
        acts, _ = self.gen.forward(G.concat_cols(z, c))
        return acts

    # ---- stream placement (round 5) ---------------------------------------------------------------------------------------
    # The HIP runtime deals every stream to one of its 4 hardware queues (least referenced queue at creation; torch hands out
    # its 32 pooled streams in order), and streams on one queue serialise.  WHICH of the step's eight side streams share a
    # queue -- with each other and with the critical chain's -- therefore follows from how many streams the process created
    # before: measured 12.4 ... 14.8 ms per step for the same launches (tools/gan_streams_probe.py: period 4 in the number of
    # earlier streams, worst when the d-loss chain lands on the critical chain's queue).  HIP has no call to place a stream,
    # so the first step TRIES: eight candidate stream sets, each one pool position further than the last, are timed on
    # `apply=False` steps (no update: the trajectory is untouched; ~0.6 s once) and the fastest set is kept (four candidates left the
    # step at 12.6-13.2 ms in half of the cases, eight at 12.1-12.9).  CFL_GAN_TUNE_STREAMS=0: off.
    tune_streams = os.environ.get('CFL_GAN_TUNE_STREAMS', '1') not in ('0', '')

    def _stream_set(self):
        from .gan_blocks import Workspace
        if not Workspace.overlap:
            return None
        new = lambda: torch.cuda.Stream(device=self.device)
        return {'gen_side': new(), 'disc_side': new(), 'chain': [new(), new()], 'chain_side': [new(), new()],
                'prep': [new(), new()]}

    def _install_streams(self, st):
        torch.cuda.synchronize(self.device)          # nothing may be in flight on the streams that go
        self.gen.ws.side_stream, self.disc.ws.side_stream = st['gen_side'], st['disc_side']
        for k in (1, 2):
            self.disc.chain(k)                        # (creates the chain workspaces on first use)
        self.disc._chains = [(st['chain'][k], self.disc._chains[k][1]) for k in range(2)]
        for k in range(2):
            self.disc._chains[k][1].side_stream = st['chain_side'][k]
        self.gen._prep_stream, self.disc._prep_stream = st['prep']
        for net in (self.gen, self.disc):
            net._prep_event = None

    def _tune_streams(self, run, candidates=8, steps=4):
        """run(): one non-applying step.  Returns the per-candidate step times (ms)."""
        import time
        times, sets, spacers = [], [], []
        for c in range(candidates):
            if c:
                spacers.append(torch.cuda.Stream(device=self.device))    # one pool position further (8 streams per set: 8 = 0 mod 4)
            st = self._stream_set()
            if st is None:
                return []
            self._install_streams(st)
            run()
            # the timed steps must take the form the training steps take -- gradient-penalty chain forked at the top of the
            # step -- which needs every cache marked as prepared ahead (nothing is rebuilt here: the weights have not moved)
            self.disc.prepare_caches(); self.gen.prepare_caches()
            run()
            torch.cuda.synchronize(self.device)
            t0 = time.perf_counter()
            for _ in range(steps):
                run()
            torch.cuda.synchronize(self.device)
            times.append((time.perf_counter() - t0) / steps * 1e3)
            sets.append(st)
        self._install_streams(sets[int(np.argmin(times))])
        self.disc.prepare_caches(); self.gen.prepare_caches()
        self.stream_tuning = [round(t, 3) for t in times]
        return times

    def step(self, real, enc_act, prj_c, neg_c, neg_tgt_act, z, eps, apply=True, x_hat=None):
        """One post-epoch iteration.  All arguments are fp32 device tensors:
             real [B, prod(ae_shape)]  ae-normalised unlabeled target images
             enc_act, prj_c, neg_c, neg_tgt_act [B, L]  encoder-side constants (see module docstring)
             z [B, z_dim], eps [B, 1]
             x_hat [B, prod(ae_shape)] or None: the perturbed images, when the caller has them (data parallelism: computed from
             the global batch); default: real + lambda_dra * std(real) * eps of THIS batch
           Returns the scalars tensor (a view; read it after the step)."""
        B, Ld, sc = self.B, self.latent_size, self.scalars
        gen, disc = self.gen, self.disc
        if GanPhase.tune_streams and not getattr(self, '_streams_tuned', False):
            self._streams_tuned = True
            self._tune_streams(lambda: self.step(real, enc_act, prj_c, neg_c, neg_tgt_act, z, eps, apply=False, x_hat=x_hat))
        marks = [] if self.timing is not None else None     # (diagnostic: CFL-event marks along the chains, tools/gan_chain_probe.py)

        def mark(name, stream=None):
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(stream if stream is not None else torch.cuda.current_stream())
                marks.append((name, ev))
        mark('start')
        main = torch.cuda.current_stream()
        c1, c2 = disc.chain(1), (disc.chain(2) if self.lambda_gp else None)
        # The gradient-penalty chain -- X_hat, its OWN discriminator forward, the backward to the input and the double
        # backward -- depends on nothing the generator produces: it starts HERE, on its own stream and workspace, beside the
        # generator forward (2 ms of layers far too small to fill 256 CUs) instead of behind the batched discriminator
        # forward, which then carries 4 B rows instead of 5 B.  (Round 5; CFL_GAN_GP_EARLY=0: X_hat rides in the batched forward)
        # ... but only when every discriminator layer's cache (weight-norm scale, filter planes) was prepared ahead for the
        # current weights: the caches' validity bits are HOST state shared by all streams, so a cache built lazily by this
        # chain's forward would be read by the main stream's forward with no dependency on the kernels that fill it (first
        # step after construction / load_state, CFL_GAN_PREP_AHEAD=0: X_hat rides in the batched forward for that step)
        gp_early = bool(self.lambda_gp) and c2 is not None and GanPhase.gp_early and disc.caches_prepared()
        if gp_early:
            start_ev = torch.cuda.Event()
            start_ev.record(main)

        def gp_chain():
            # (enqueued BEHIND the generator forward -- the host needs ~3 ms to enqueue this chain, and the critical chain's
            # first launches must not wait for that -- but ordered only behind the start of the step)
            c2[0].wait_event(start_ev)
            with torch.cuda.stream(c2[0]):
                xh = G.perturb(real, eps, self.lambda_dra) if x_hat is None else x_hat
                _, _, tape = disc.forward(xh, ws=c2[1])
                disc.gp_grads(tape, 0, B, self.lambda_gp, sc[S_D_GP:S_D_GP + 1], disc.pool.grad2, ws=c2[1])
                mark('chain2_gradient_penalty')
                return tape if self.keep_tapes else None
        # ---- generator: rows [g | g_prj | g_neg] ------------------------------------------------
        zc = torch.empty(3 * B, self.z_dim + Ld, dtype=torch.float32, device=self.device)
        for i, c in enumerate((enc_act, prj_c, neg_c)):
            G.concat_cols(z, c, out=zc[i * B:(i + 1) * B])
        fake, g_tape = gen.forward(zc)
        mark('g_forward')
        gp_tape = gp_chain() if gp_early else None
        # ---- discriminator: rows [real | g | g_prj | g_neg | X_hat] -----------------------------
        nrow = 5 * B if (self.lambda_gp and not gp_early) else 4 * B
        x_all = torch.empty(nrow, self.ae_size, dtype=torch.float32, device=self.device)
        x_all[:B].copy_(real)
        x_all[B:4 * B].copy_(fake)
        if self.lambda_gp and not gp_early:
            if x_hat is None:
                G.perturb(real, eps, self.lambda_dra, out=x_all[4 * B:5 * B])
            else:
                x_all[4 * B:5 * B].copy_(x_hat)
        d_logit, d_lat, d_tape = disc.forward(x_all)
        mark('d_forward')

        # ---- discriminator loss and gradient (variables only) -----------------------------------
        dd = torch.zeros(3 * B, 1, dtype=torch.float32, device=self.device)
        dl = torch.zeros(3 * B, Ld, dtype=torch.float32, device=self.device)
        G.bce_logits(d_logit[0:B], 1.0, 1.0, sc[S_D_REAL:S_D_REAL + 1], sc[S_FRAC_REAL:S_FRAC_REAL + 1], dd[0:B])
        G.bce_logits(d_logit[B:2 * B], 0.0, 0.5, sc[S_D_ENC:S_D_ENC + 1], sc[S_FRAC_FAKE:S_FRAC_FAKE + 1],
                     dd[B:2 * B])
        G.bce_logits(d_logit[2 * B:3 * B], 0.0, 0.5, sc[S_D_PRJ:S_D_PRJ + 1], sc[S_FRAC_PRJ:S_FRAC_PRJ + 1],
                     dd[2 * B:3 * B])
        G.rowdist_loss(d_lat[0:B], enc_act, 0, 0.0, 1.0, sc[S_D_LAT:S_D_LAT + 1], dl[0:B])
        # Three backward chains hang off the one discriminator forward and do not feed each other: the d-loss backward
        # (discriminator variables), the gradient-penalty passes (X_hat rows, into grad2) and the g-loss backward through D
        # and then G.  Most of their ~500 launches are far too small to fill 256 CUs, so the first two run on streams of
        # their own (own workspaces) beside the third; they are joined before the gradients are added and applied.
        if c1 is not None:
            c1[0].wait_stream(main)
            with torch.cuda.stream(c1[0]):
                disc.backward(d_tape, 0, 3 * B, dd, dl, need_dx=False, need_dw=True, grad=disc.pool.grad, ws=c1[1])
                mark('chain1_d_loss_backward')
        else:
            disc.backward(d_tape, 0, 3 * B, dd, dl, need_dx=False, need_dw=True, grad=disc.pool.grad)
        if self.lambda_gp and not gp_early:
            if c2 is not None:
                c2[0].wait_stream(main)
                with torch.cuda.stream(c2[0]):
                    disc.gp_grads(d_tape, 4 * B, 5 * B, self.lambda_gp, sc[S_D_GP:S_D_GP + 1], disc.pool.grad2, ws=c2[1])
                    mark('chain2_gradient_penalty')
            else:
                disc.gp_grads(d_tape, 4 * B, 5 * B, self.lambda_gp, sc[S_D_GP:S_D_GP + 1], disc.pool.grad2)

        # ---- generator loss: gradient w.r.t. the images [g | g_prj | g_neg], then through G ------
        gd = torch.zeros(3 * B, 1, dtype=torch.float32, device=self.device)
        gl = torch.zeros(3 * B, Ld, dtype=torch.float32, device=self.device)
        G.bce_logits(d_logit[B:2 * B], 1.0, 0.5, sc[S_G_ENC:S_G_ENC + 1], None, gd[0:B])
        G.bce_logits(d_logit[2 * B:3 * B], 1.0, 0.5, sc[S_G_PRJ:S_G_PRJ + 1], None, gd[B:2 * B])
        if self.m_enc:
            G.rowdist_loss(d_lat[B:2 * B], enc_act, 1, self.m_enc, 1.0, sc[S_G_LAT:S_G_LAT + 1], gl[0:B])
        else:
            G.rowdist_loss(d_lat[B:2 * B], enc_act, 0, 0.0, 1.0, sc[S_G_LAT:S_G_LAT + 1], gl[0:B])
        if self.m_prj:
            G.rowdist_loss(d_lat[3 * B:4 * B], neg_tgt_act, 2, self.m_prj, 1.0, sc[S_G_NEG:S_G_NEG + 1],
                           gl[2 * B:3 * B])
        # (tried: this chain on a high-priority stream -- 15.5 -> 16.4 ms, tools/gan_chain_probe.py; not kept)
        d_img = disc.backward(d_tape, B, 4 * B, gd, gl, need_dx=True, need_dw=False)
        mark('main_g_loss_backward_through_d')
        gen.backward(g_tape, d_img.contiguous())
        mark('main_g_backward')
        for c in (c1, c2):
            if c is not None:
                main.wait_stream(c[0])
        if self.lambda_gp:
            G.axpy(1.0, disc.pool.grad2, disc.pool.grad)
        if self.keep_tapes:     # (X_hat's activations: rows [4B, 5B) of d_tape, or rows [0, B) of the third tape)
            self.last_tapes = (g_tape, d_tape, gp_tape if gp_early else None)

        if apply:
            scale = self._exchange()
            disc.adam(scale)
            gen.adam(scale)
        mark('end')
        if marks is not None:
            torch.cuda.synchronize()
            t0 = marks[0][1]
            self.timing = {name: t0.elapsed_time(ev) for name, ev in marks}
        return sc

    def step_cgan(self, real_pos, real_neg, pos_c, neg_c, z, eps, apply=True, x_hat=None):
        """One post-epoch iteration of the --cgan branch.  real_pos / real_neg: ae-normalised positive / negative
        TARGET images [B, prod(ae_shape)]; pos_c / neg_c [B, c_dim]: the conditions (source side)."""
        B, sc, hb = self.B, self.scalars, self.B // 2
        gen, disc = self.gen, self.disc
        cd = self.c_dim
        # conditions of the rows [real | g | neg | int | X_hat]
        half = pos_c[:hb].clone()
        G.axpy(1.0, pos_c[hb:], half)
        half.mul_(0.5)              # (c[:B/2] + c[B/2:]) / 2  (in-place scale of a scratch tensor: plumbing)
        nrow = 3 * B + hb + (B if self.lambda_gp else 0)
        t_all = torch.empty(nrow, cd, dtype=torch.float32, device=self.device)
        t_all[0:B].copy_(pos_c)
        t_all[B:2 * B].copy_(pos_c)
        t_all[2 * B:3 * B].copy_(neg_c)
        t_all[3 * B:3 * B + hb].copy_(half)
        if self.lambda_gp:
            t_all[3 * B + hb:].copy_(pos_c)
        # generator rows [g | g_int]
        zc = torch.empty(B + hb, self.z_dim + cd, dtype=torch.float32, device=self.device)
        G.concat_cols(z, pos_c, out=zc[:B])
        G.concat_cols(z[:hb], half, out=zc[B:])
        fake, g_tape = gen.forward(zc)
        x_all = torch.empty(nrow, self.ae_size, dtype=torch.float32, device=self.device)
        x_all[0:B].copy_(real_pos)
        x_all[B:2 * B].copy_(fake[:B])
        x_all[2 * B:3 * B].copy_(real_neg)
        x_all[3 * B:3 * B + hb].copy_(fake[B:])
        if self.lambda_gp:
            if x_hat is None:
                G.perturb(real_pos, eps, self.lambda_dra, out=x_all[3 * B + hb:])
            else:
                x_all[3 * B + hb:].copy_(x_hat)
        d_logit, _, d_tape = disc.forward(x_all, t_all)

        dd = torch.zeros(3 * B, 1, dtype=torch.float32, device=self.device)
        G.bce_logits(d_logit[0:B], 1.0, 1.0, sc[S_D_REAL:S_D_REAL + 1], sc[S_FRAC_REAL:S_FRAC_REAL + 1], dd[0:B])
        G.bce_logits(d_logit[B:2 * B], 0.0, 0.5, sc[S_D_ENC:S_D_ENC + 1], sc[S_FRAC_FAKE:S_FRAC_FAKE + 1], dd[B:2 * B])
        G.bce_logits(d_logit[2 * B:3 * B], 0.0, 0.5, sc[S_D_PRJ:S_D_PRJ + 1], None, dd[2 * B:3 * B])
        disc.backward(d_tape, 0, 3 * B, dd, None, need_dx=False, need_dw=True, grad=disc.pool.grad)
        if self.lambda_gp:
            disc.gp_grads(d_tape, 3 * B + hb, nrow, self.lambda_gp, sc[S_D_GP:S_D_GP + 1], disc.pool.grad2)
            G.axpy(1.0, disc.pool.grad2, disc.pool.grad)

        gd1 = torch.zeros(B, 1, dtype=torch.float32, device=self.device)
        gd2 = torch.zeros(hb, 1, dtype=torch.float32, device=self.device)
        G.bce_logits(d_logit[B:2 * B], 1.0, 1.0, sc[S_G_ENC:S_G_ENC + 1], None, gd1)
        G.bce_logits(d_logit[3 * B:3 * B + hb], 1.0, 1.0, sc[S_G_PRJ:S_G_PRJ + 1], None, gd2)
        d_img = torch.empty(B + hb, self.ae_size, dtype=torch.float32, device=self.device)
        d_img[:B].copy_(disc.backward(d_tape, B, 2 * B, gd1, None, need_dx=True, need_dw=False))
        d_img[B:].copy_(disc.backward(d_tape, 3 * B, 3 * B + hb, gd2, None, need_dx=True, need_dw=False))
        gen.backward(g_tape, d_img)
        if apply:
            scale = self._exchange()
            disc.adam(scale)
            gen.adam(scale)
        return sc

    def read_scalars(self):
        s = self.scalars.detach().cpu().numpy()
        if self.cgan:
            out = {'d_loss_real': float(s[S_D_REAL]), 'd_loss_fake': float(2.0 * s[S_D_ENC]),
                   'd_loss_neg': float(2.0 * s[S_D_PRJ]), 'd_grad_loss': float(s[S_D_GP]) if self.lambda_gp else 0.0,
                   'g_loss': float(s[S_G_ENC]), 'g_loss_int': float(s[S_G_PRJ]),
                   'd_real_accuracy': float(s[S_FRAC_REAL]), 'd_fake_accuracy': float(1.0 - s[S_FRAC_FAKE]),
                   'g_accuracy': float(s[S_FRAC_FAKE]), 'd_loss_d': 0.0, 'g_loss_d': 0.0, 'g_loss_d_neg': 0.0}
            out['d_total_loss'] = out['d_loss_real'] + 0.5 * (out['d_loss_fake'] + out['d_loss_neg']) + out['d_grad_loss']
            out['g_total_loss'] = out['g_loss'] + out['g_loss_int']
            return out
        out = {
            'd_loss_real': float(s[S_D_REAL]), 'd_loss_fake': float(s[S_D_ENC] + s[S_D_PRJ]),
            'd_grad_loss': float(s[S_D_GP]) if self.lambda_gp else 0.0, 'd_loss_d': float(s[S_D_LAT]),
            'g_loss': float(s[S_G_ENC] + s[S_G_PRJ]), 'g_loss_d': float(s[S_G_LAT]),
            'g_loss_d_neg': float(s[S_G_NEG]) if self.m_prj else 0.0,
            'd_real_accuracy': float(s[S_FRAC_REAL]), 'd_fake_accuracy': float(1.0 - s[S_FRAC_FAKE]),
            'g_accuracy': float(s[S_FRAC_FAKE]),
        }
        out['d_total_loss'] = out['d_loss_real'] + out['d_loss_fake'] + out['d_grad_loss'] + out['d_loss_d']
        out['g_total_loss'] = out['g_loss'] + out['g_loss_d'] + out['g_loss_d_neg']
        return out

    def state(self):
        return {'generator': self.gen.state(), 'discriminator': self.disc.state()}

    def load_state(self, st):
        self.gen.load_state(st['generator'])
        self.disc.load_state(st['discriminator'])
