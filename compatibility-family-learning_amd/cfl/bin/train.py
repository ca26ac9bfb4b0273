"""python -m cfl.bin.train -- CFL training: distance epochs (``--model-type linear|conv``) followed,
with ``--gan``, by the MrCGAN post epochs.

Drop-in for the reference's cfl/bin/train.py flag surface and directory layout
(checkpoints/<data>/<model.get_name()>/{model-*, best_model/, best_acc_model/}).
Random crop / mirror / resize input transformers run on the GPU (cfl_image_transform)."""
import logging
import os

from .. import engine as dp
from ..input_data import load_data_sets
from ..models.cfl import construct_model
from ..ops import dist_ae_transformer, dist_normalizer, dist_transformer
from ..utils import Saver, ScalarWriter, dist_check_args, dist_parser, load_model, reduce_product
from .train_dist import setup_logging

logger = logging.getLogger(__name__)


def train_dist(load_pre_weights, data_switch, epochs, post_epochs, eval_epochs, save_iters,
               disable_eval, reset, data_name, data_root, checkpoint_root, log_root, seed,
               data_mirror, data_random_crop, data_is_image, raw_latent, data_scale, data_mean,
               latent_norm, **model_args):
    a = model_args
    dp.init_from_env()
    chief = dp.rank() == 0
    input_size = reduce_product(a['input_shape'])
    source_size = reduce_product(a['source_shape']) if a['source_shape'] else input_size
    data = load_data_sets(os.path.join(data_root, data_name), source_size, is_image=data_is_image,
                          is_double=a['data_is_double'], raw_latent=raw_latent,
                          directed=a['directed'] or a['data_directed'], data_switch=data_switch, seed=seed)
    (data_normalizer, data_unnormalizer, ae_normalizer, ae_unnormalizer,
     latent_normalizer) = dist_normalizer(
        input_shape=a['input_shape'], ae_shape=a['ae_shape'], data_scale=data_scale,
        data_mean=data_mean, data_norm=a['data_norm'], latent_norm=latent_norm, data_type=a['data_type'])
    train_tr, val_tr = dist_transformer(a['source_shape'], a['input_shape'], data_random_crop, data_mirror)
    model, aux = construct_model(
        train_data_transformer=train_tr, val_data_transformer=val_tr,
        ae_transformer=dist_ae_transformer(a['input_shape'], a['ae_shape']),
        is_double=a.pop('data_is_double'), disable_double=a.pop('data_disable_double'), data=data,
        data_normalizer=data_normalizer, data_unnormalizer=data_unnormalizer,
        ae_normalizer=ae_normalizer, ae_unnormalizer=ae_unnormalizer,
        latent_normalizer=latent_normalizer, enable_input_producer=True, seed=seed, **a)

    root = os.path.join(checkpoint_root, data_name)
    no_gan_checkpoint_dir = os.path.join(root, model.get_name(no_gan=True)) if load_pre_weights else None
    checkpoint_dir = os.path.join(root, model.get_name())
    log_dir = os.path.join(log_root, data_name, model.get_name())
    dp.prepare_run_dirs((checkpoint_dir, log_dir), reset)
    setup_logging(log_dir)
    logger.warning('run with %s', model.get_name())

    saver, start_iter = load_model(model, checkpoint_dir, no_gan_checkpoint_dir)
    best_dir = os.path.join(checkpoint_dir, 'best_model')
    best_acc_dir = os.path.join(checkpoint_dir, 'best_acc_model')
    os.makedirs(best_dir, exist_ok=True)
    os.makedirs(best_acc_dir, exist_ok=True)
    writer = ScalarWriter(log_dir) if chief else None      # checkpoints, scalars and evaluation: rank 0 only
    try:
        model.train(sess=None, data=data, start_iter=start_iter, epochs=epochs, post_epochs=post_epochs,
                    best_dir=best_dir, best_acc_dir=best_acc_dir, checkpoint_dir=checkpoint_dir,
                    eval_epochs=eval_epochs, disable_eval=disable_eval, saver=saver, best_saver=Saver(),
                    best_acc_saver=Saver(), save_iters=save_iters, writer=writer)
    finally:
        if writer is not None:
            writer.close()


def parse_args(argv=None):
    parser = dist_parser(batch_size=100)
    parser.add_argument('--load-pre-weights', action='store_true')
    parser.add_argument('--epochs', type=int, default=120)
    parser.add_argument('--save-iters', type=int)
    parser.add_argument('--data-switch', action='store_true')
    parser.add_argument('--post-epochs', type=int, default=100)
    parser.add_argument('--eval-epochs', type=int, default=1)
    parser.add_argument('--disable-eval', action='store_true')
    parser.add_argument('--reset', action='store_true')
    args = parser.parse_args(argv)
    dist_check_args(args)
    return args


def main(argv=None):
    try:
        train_dist(**vars(parse_args(argv)))
    finally:
        if dp.world_size() > 1:
            dp.finalize()       # the raw RCCL communicator of the exchange, then the process group (every rank)


if __name__ == '__main__':
    main()
