"""python -m cfl.bin.train_dist -- train the Monomer-data model on MI355X.

Drop-in for the reference's cfl/bin/train_dist.py (same flags, checkpoint / log
layout and best_acc_model bookkeeping, cfl/bin/train_dist.py:37-124).  The
TensorFlow queues + enqueue threads are replaced by HBM-resident feature tables
(cfl.input_data.ResidentFeatures) that the fused HIP training step reads in place
by pair index (cfl_pair_train_step_idx: no gather pass, no batch copy), and the
per-iteration ``sess.run`` by that one step.
"""
import gc
import logging
import os

import numpy as np

from tqdm import trange

from .. import engine as dp
from ..input_data import feature_source, load_data_sets
from ..models.dist import construct_model
from ..ops import normalizer, unnormalizer
from ..utils import (IncrementalAverage, Saver, dist_eval, load_best_stats, load_model, log_args,
                     monomer_parser, reduce_product, save_best_stats)

logger = logging.getLogger(__name__)

SCALAR_EVERY = 25  # default host read-back cadence of the display scalars and the validation fetch (steps); --scalar-every N
                   # changes it, N = 1 is the reference's cadence (one val_s_accuracy fetch per iteration,
                   # cfl/bin/train_dist.py:79-86: the validation SemiDataSet then advances exactly as the reference's does)


class DeferredScalars(object):
    """Read-backs that do not stall the stream: at a read-back point the step's scalars and the validation accuracy
    (computed on the device) are copied into pinned host memory behind an event; they are handed to the callback
    when the event has completed -- normally one cadence later -- and all of them at `flush()`.  The host can thus
    stay hundreds of launches ahead of the GPU, which is what hides the per-epoch reshuffle of the pair lists
    (a serial MT19937 permutation, cfl/input_data.py:543-551) behind queued training steps."""

    SLOTS = 64      # pinned host slots (allocated once per model: hipHostMalloc is slow); more than enough run-ahead

    def __init__(self, model, on_scalars):
        self.model, self.on_scalars, self.pending = model, on_scalars, []
        self.pool, self.next_slot = None, 0

    def _slot(self, width):
        """Pinned row [scalars | validation scores of the positive pairs | ... of the negative pairs]."""
        import torch
        pool = getattr(self.model, '_deferred_pool', None)
        if pool is None or pool.shape[1] != width:
            if self.pending:
                self.poll(wait=True)
            pool = self.model._deferred_pool = torch.empty(self.SLOTS, width, dtype=torch.float32).pin_memory()
        self.pool = pool
        host = pool[self.next_slot]
        self.next_slot = (self.next_slot + 1) % self.SLOTS
        return host

    def reserve(self, n, val_rows):
        """`n` pinned slot rows for a fused train + validation call (the kernels write [scalars | scores] into them
        themselves: PairEngine.step_windows_val); waits for old read-backs when the ring is full"""
        from .. import hipabi as H
        while len(self.pending) + n > self.SLOTS:
            self.poll(wait=True, at_most=1)
        return [self._slot(H.S_COUNT + 2 * val_rows) for _ in range(n)]

    def commit(self, steps, slots, val_rows):
        """the call that fills `slots` (one per entry of `steps`) has been enqueued: ONE event behind it for all of them"""
        import torch
        ev = torch.cuda.Event()
        ev.record()
        eng = self.model.engine
        for step, host in zip(steps, slots):
            self.pending.append((step, host, ev, eng._scalar_scale, val_rows))
        self.poll()

    def record(self, step, val_batch):
        self.record_scalars(step, *self.record_scores(val_batch, scalars_too=True))

    def record_scores(self, val_batch, scalars_too=False):
        """score a validation batch with the weights as they stand NOW into a fresh slot; (slot, pairs per group)"""
        from .. import hipabi as H
        eng = self.model.engine
        while len(self.pending) >= self.SLOTS:
            self.poll(wait=True, at_most=1)
        table, streams = val_batch
        n = streams.n
        host = self._slot(H.S_COUNT + 2 * n)
        if scalars_too:
            host[:H.S_COUNT].copy_(eng.scalars, non_blocking=True)
        # the validation scores travel as they are (2 x n floats); the accuracy is counted on the host when the
        # slot is read -- no elementwise / reduction launches on the training stream
        # (positive and negative pairs in ONE scoring call: two launches and one copy instead of four and two.  Round 4 tried
        # a device staging row + ONE copy on a copy stream behind an event instead of the two copies on the training stream:
        # 39.1 -> 42.2 us per step, the cross-stream event costs more than the second copy; not kept)
        host[H.S_COUNT:].copy_(eng.scores_pos_neg(table, streams), non_blocking=True)
        return host, n, scalars_too

    def record_scalars(self, step, host, n, have_scalars=False):
        """... and the scalars of the step that has just been enqueued; one event behind both"""
        import torch
        from .. import hipabi as H
        eng = self.model.engine
        if not have_scalars:
            host[:H.S_COUNT].copy_(eng.scalars, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((step, host, ev, eng._scalar_scale, n))
        self.poll()

    def poll(self, wait=False, at_most=None):
        """hand completed read-backs to the callback.  Entries that share an event (the up to FUSED_CHUNK read-backs of one
        fused library call) are turned into scalars and accuracies TOGETHER, with a handful of NumPy calls for all of them:
        at --scalar-every 1 the per-entry Python of the previous form (a ctypes status call, a dict and four small NumPy
        reductions per iteration: ~25 us) was what bounded the loop, not the GPU."""
        from .. import hipabi as H
        k = 0
        while self.pending and (wait or self.pending[0][2].query()) and (at_most is None or k < at_most):
            ev = self.pending[0][2]
            ev.synchronize()
            group = []
            while self.pending and self.pending[0][2] is ev and (at_most is None or k < at_most):
                group.append(self.pending.pop(0))
                k += 1
            n, scale = group[0][4], group[0][3]
            rows = np.stack([g[1].numpy() for g in group]) if len(group) > 1 else group[0][1].numpy()[None]
            sc = rows[:, :H.S_COUNT]
            # a lost in-launch hand-off (sticky error word) raises here, one cadence after it happened at the latest
            # (the sticky error word is compared as the float it is: 0.0 healthy, anything else -- 1.0, a sum of them, NaN -- not)
            if (sc[:, H.S_ERROR] != 0).any() or getattr(self.model.engine, '_oneshot', None) is not None:
                for r in sc:
                    self.model.engine.check_health(r)
            # batch_accuracy (cfl/bin/train_dist.py:52-56 of the reference: mean of [s_pos > 0] and [s_neg <= 0])
            acc = (0.5 / n) * ((rows[:, H.S_COUNT:H.S_COUNT + n] > 0).sum(1) + (rows[:, H.S_COUNT + n:] <= 0).sum(1))
            vals = (sc.astype(np.float64) * scale).tolist() if scale != 1.0 else sc.tolist()
            for g, row, a in zip(group, vals, acc.tolist()):
                self.on_scalars(g[0], dict(zip(H.SCALAR_NAMES, row)), a)

    def flush(self):
        self.poll(wait=True)


FUSED_CHUNK = 16    # iterations per library call at --scalar-every 1 (4 such calls fit the 64-slot pinned ring)


def H_fusable(eng, batch_size, rows=None):
    """can a training step of `rows` rows per group (a rank's shard; default the whole batch) carry a validation batch of
    `batch_size` pairs per group inside its own launches?"""
    from .. import hipabi as H
    if not hasattr(eng, 'step_windows_val'):
        return False
    return H.train_val_fusable(eng.shape, rows if rows is not None else batch_size, batch_size)


def train_steps(model, train_src, val_src, batch_size, shard, n_steps, on_scalars=None, progress=None,
                scalar_every=SCALAR_EVERY):
    """The iterations of the training loop (cfl/bin/train_dist.py:77-87 of the reference: one
    ``sess.run([summary, [s_optim], s_accuracy, val_s_accuracy])`` per iteration): the labeled batches of the
    seeded index stream -- as positions into the resident feature table -- go through the fused training step;
    the display scalars (train / validation accuracy) are read back every `scalar_every` iterations, without stalling
    the stream (DeferredScalars).  The iterations between two read-backs are one engine call over windows of the
    device pair lists (on one GPU one library call; data parallel: the same windows with the gradient exchange inside
    the loop, PairEngine.step_windows)."""
    i = 0
    every = max(1, int(scalar_every))
    deferred = DeferredScalars(model, on_scalars) if on_scalars is not None else None
    # The read-back iterations carry their validation batch INSIDE the training step's launches (extra scoring rows,
    # cfl_pair_train_val_steps_idx_planes) wherever the library can (single GPU, chunk-at-a-time projection + wave-per-row
    # math: the batch sizes of this loop): no second projection / row-math launch pair, no copy commands -- the kernels write
    # [scalars | validation scores] into the pinned ring themselves.  With --scalar-every 1 (the reference's cadence) up to
    # FUSED_CHUNK iterations go into one library call.  CFL_FUSED_VAL=0: the separate scoring call of rounds 3-4.
    eng = model.engine
    # (data parallel, round 6: the same fused loop when the library drives the exchange itself -- PairEngine.dp_native: RCCL through
    # the raw communicator or the one-shot exchange; every rank carries the whole validation batch, theta being replicated)
    dp_ok = (not dp.dp_active()) or (hasattr(eng, 'dp_native') and eng.dp_native() is not None)
    rows = (shard[1] - shard[0]) if shard is not None else batch_size
    fused = (deferred is not None and dp_ok and os.environ.get('CFL_FUSED_VAL', '1') not in ('0', '')
             and (shard is None or dp.dp_active()) and H_fusable(eng, batch_size, rows))
    while fused and i < n_steps:
        is_readback = lambda k: k % every == 0 or k == n_steps - 1
        # chunk [i, j): with a read-back every iteration, FUSED_CHUNK of them; otherwise up to and including the next one
        j = min(n_steps, i + FUSED_CHUNK) if every == 1 else min((i + every - 1) // every * every, n_steps - 1) + 1
        kt, kv = train_src.available_windows(batch_size), val_src.available_windows(batch_size)
        j = min(j, i + kt)
        mask = [is_readback(k) for k in range(i, j)]
        while sum(mask) > kv and mask:      # not enough plain validation windows left (the validation list wraps): a shorter chunk
            mask.pop()
        j = i + len(mask)
        if j <= i or (sum(mask) == 0 and every == 1):
            # the very next training or validation batch wraps an epoch (its list is reshuffled first): ONE iteration through
            # the same fused call, on the windows the general draw leaves; batches that are no windows at all (B larger
            # than a pair list: oversampled) go the general way, scored with the weights BEFORE the update as well
            rb = is_readback(i)
            win1 = train_src.next_window_any(batch_size, shard)
            if win1 is None:
                slot = deferred.record_scores(val_src.next_indexed(batch_size)) if rb else None
                eng.step(train_src.next_indexed(batch_size, shard))
                if slot is not None:
                    deferred.record_scalars(i, slot[0], slot[1])
            elif not rb:
                eng.step_windows(win1)
            else:
                vwin1 = val_src.next_window_any(batch_size)
                if vwin1 is None:
                    slot = deferred.record_scores(val_src.next_indexed(batch_size))
                    eng.step_windows(win1)
                    deferred.record_scalars(i, slot[0], slot[1])
                else:
                    slots = deferred.reserve(1, batch_size)
                    eng.step_windows_val(win1, vwin1, [True], [slots[0].data_ptr()])
                    deferred.commit([i], slots, batch_size)
            i += 1
            if progress is not None:
                progress.update(1)
            continue
        win = train_src.next_windows(batch_size, j - i, shard)
        nval = sum(mask)
        if nval:
            vwin = val_src.next_windows(batch_size, nval)
            slots = deferred.reserve(nval, batch_size)
            eng.step_windows_val(win, vwin, mask, [h.data_ptr() for h in slots])
            deferred.commit([i + k for k, mk in enumerate(mask) if mk], slots, batch_size)
        else:
            eng.step_windows(win)
        if progress is not None:
            progress.update(j - i)
        i = j
    while i < n_steps:
        # the chunk ends with the next iteration whose scalars are read back (0, 25, 50, ..., and the last one).  The validation
        # batch of a read-back iteration is scored with the weights BEFORE that iteration's update, as the fused loop above does
        # and as one sess.run of the reference fetches it (cfl/bin/train_dist.py:81-82): the iterations in front of it go first,
        # then the scoring call, then the read-back iteration itself
        stop = min((i + every - 1) // every * every, n_steps - 1) + 1
        is_rb = lambda k: k % every == 0 or k == n_steps - 1
        ahead = stop - i - 1 if (deferred is not None and is_rb(stop - 1)) else stop - i
        win = train_src.next_windows(batch_size, ahead, shard) if ahead > 0 else None
        if win is not None:
            model.engine.step_windows(win)
            done = win.nsteps
        else:
            last = i
            slot = deferred.record_scores(val_src.next_indexed(batch_size)) if (deferred is not None and is_rb(last)) else None
            win1 = train_src.next_windows(batch_size, 1, shard)
            if win1 is not None:
                model.engine.step_windows(win1)
            else:
                model.engine.step(train_src.next_indexed(batch_size, shard))
            if slot is not None:
                deferred.record_scalars(last, slot[0], slot[1])
            done = 1
        i += done
        if progress is not None:
            progress.update(done)
    if deferred is not None:
        deferred.flush()


def train_loop(model, data, aux, batch_size, start_epoch, epochs, log_dir, checkpoint_dir, saver,
               scalar_every=SCALAR_EVERY):
    best_saver = Saver()
    nb_train = max(data.train.num_examples_labeled_pos, data.train.num_examples_labeled_neg)
    logger.warning('%d examples', nb_train)
    logger.warning('model: %s', model.get_name())
    nb_batch = nb_train // batch_size

    train_src = feature_source(aux.train, model.device)     # resident in HBM, or streamed when the table does not fit
    val_src = feature_source(aux.val, model.device)

    best_dir = os.path.join(checkpoint_dir, 'best_acc_model')
    os.makedirs(best_dir, exist_ok=True)
    best_accuracy_path = os.path.join(best_dir, 'best_accuracy')
    stats = load_best_stats(best_accuracy_path)
    # data parallel (torchrun): every rank walks the same seeded index stream and
    # trains on its contiguous slice of each global batch; rank 0 evaluates and saves
    shard = dp.shard_rows(batch_size) if dp.world_size() > 1 else None
    chief = dp.rank() == 0
    # the scalars are global-batch values on every rank (they travel in the gradient all-reduce): one writer
    scalar_log = open(os.path.join(log_dir, 'scalars.tsv'), 'a') if chief else None
    for e in range(start_epoch, epochs):
        t = trange(nb_batch, disable=not chief)
        t.set_description('epoch {}'.format(e))
        train_avg, val_avg = IncrementalAverage(), IncrementalAverage()

        bad = []

        def on_scalars(i, s, val_acc, e=e, t=t, train_avg=train_avg, val_avg=val_avg, bad=bad):
            if not np.isfinite(s['total']):
                bad.append(nb_batch * e + i)
            train_avg.add(s['accuracy'])
            val_avg.add(val_acc)
            t.set_postfix(train_acc=train_avg.average, val_acc=val_avg.average)
            if scalar_log is not None:
                scalar_log.write('{}\t{}\t{}\t{}\n'.format(nb_batch * e + i, s['total'], s['accuracy'],
                                                           s['threshold']))
        train_steps(model, train_src, val_src, batch_size, shard, nb_batch, on_scalars, progress=t,
                    scalar_every=scalar_every)
        t.close()
        if scalar_log is not None:
            scalar_log.flush()
        gc.collect()
        if bad:
            # never checkpoint poisoned parameters: the previous epoch's files stay the latest ones
            raise FloatingPointError('non-finite training loss at iteration {} (epoch {}): checkpoint not written'
                                     .format(bad[0], e))
        model.engine.sync_state()      # (collective when the Adam slots are sharded: one-shot exchange)
        if chief:
            saver.save(model, os.path.join(checkpoint_dir, 'model'), global_step=e)
        if dp.world_size() > 1:
            # the chief-only save must not let the other ranks run ahead into the next epoch's exchange unboundedly
            dp.dist.barrier()

        # evaluation is collective under data parallelism (every rank scores a shard, rank 0 gathers); the two
        # numbers come back to every rank, so all ranks take the same decisions and only rank 0 writes files
        val_stats = dist_eval(None, model, batch_size, data.val)
        if val_stats.accuracy > stats.best_accuracy:
            test_stats = dist_eval(None, model, batch_size, data.test)
            logger.warning('epoch %d: current error = train: %f val: %f test: %f / auc = val: %f test: %f',
                           e, 1. - train_avg.average, 1. - val_stats.accuracy,
                           1. - test_stats.accuracy, val_stats.auc, test_stats.auc)
            stats.best_accuracy, stats.best_auc, stats.best_epoch = val_stats.accuracy, val_stats.auc, e
            model.engine.sync_state()
            if chief:
                best_saver.save(model, os.path.join(best_dir, 'model'), global_step=stats.best_epoch)
                save_best_stats(best_accuracy_path, stats.best_epoch, stats.best_accuracy, stats.best_auc)
        else:
            logger.warning('epoch %d: avg error = train: %f val: %f', e, 1. - train_avg.average,
                           1. - val_avg.average)
    if scalar_log is not None:
        scalar_log.close()


def setup_logging(log_dir):
    log_format = '%(asctime)s [%(levelname)-5.5s] [%(name)s]  %(message)s'
    # one writer of log.log: rank 0 (the other ranks log to the console only)
    logging.basicConfig(filename=os.path.join(log_dir, 'log.log') if dp.rank() == 0 else None, format=log_format,
                        level=logging.WARNING)
    console = logging.StreamHandler()
    console.setLevel(logging.INFO)
    console.setFormatter(logging.Formatter(log_format))
    logging.getLogger().addHandler(console)


def train_monomer(data_name, data_root, checkpoint_root, log_root, run_tag, seed, normalize_value,
                  input_shape, batch_size, num_components, latent_size, lr, beta1, beta2, epochs,
                  reg_const, reset, scalar_every=SCALAR_EVERY):
    dp.init_from_env()
    input_shape = tuple(input_shape)
    input_size = reduce_product(input_shape)
    data = load_data_sets(os.path.join(data_root, data_name), input_size, seed=seed)
    model, aux = construct_model(
        input_shape=input_shape, latent_size=latent_size, normalize_value=normalize_value, lr=lr,
        beta1=beta1, beta2=beta2, num_components=num_components, batch_size=batch_size, data=data,
        run_tag=run_tag, reg_const=reg_const, data_normalizer=normalizer(normalize_value, 0., None, None),
        data_unnormalizer=unnormalizer(normalize_value, 0.), seed=seed)

    checkpoint_dir = os.path.join(checkpoint_root, data_name, model.get_name())
    log_dir = os.path.join(log_root, data_name, model.get_name())
    dp.prepare_run_dirs((checkpoint_dir, log_dir), reset)
    setup_logging(log_dir)
    saver, start_epoch = load_model(model, checkpoint_dir)
    train_loop(model=model, aux=aux, data=data, batch_size=batch_size, start_epoch=start_epoch,
               epochs=epochs, log_dir=log_dir, checkpoint_dir=checkpoint_dir, saver=saver, scalar_every=scalar_every)


def parse_args(argv=None):
    parser = monomer_parser()
    parser.add_argument('--epochs', type=int, default=120)
    parser.add_argument('--reset', action='store_true')
    # not a reference flag: the reference fetches the validation accuracy and the summaries in EVERY sess.run
    # (cfl/bin/train_dist.py:79-86); 1 reproduces that cadence (and its validation stream), the default reads back every 25
    parser.add_argument('--scalar-every', type=int, default=SCALAR_EVERY,
                        help='read the display scalars / score a validation batch every N iterations (1 = reference cadence)')
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    log_args(args)
    try:
        train_monomer(**vars(args))
    finally:
        if dp.world_size() > 1:
            dp.finalize()       # the raw RCCL communicator of the exchange, then the process group (every rank)


if __name__ == '__main__':
    main()
