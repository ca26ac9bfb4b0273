"""python -m cfl.bin.sample -- image grids from a trained MrCGAN generator (drop-in for
cfl/bin/sample.py:21-182): ``--sample-type project`` (prototype projections of the source items),
``near`` (generations conditioned on the target encoder) and ``project_disc`` (per pair, samples
of every component sorted by the discriminator).  Grids are written as PNG under
``<sample-root>/<data>/<model.get_name()>/<sample-type>/``."""
import logging
import os

from ..input_data import load_data_sets
from ..models.cfl import construct_model
from ..ops import dist_ae_transformer, dist_normalizer, dist_transformer
from ..utils import (dist_check_args, dist_parser, dist_sample, dist_sample_near,
                     dist_sample_project_disc, load_model, reduce_product)

logger = logging.getLogger(__name__)


def main(sample_root, reorder, sample_type, data_name, data_root, checkpoint_root, log_root, seed, data_mirror,
         data_random_crop, data_is_image, raw_latent, data_scale, data_mean, latent_norm, **model_args):
    a = model_args
    if not a['gan']:
        raise ValueError('cfl.bin.sample needs the generator: pass the --gan flags of the training run')
    input_size = reduce_product(a['input_shape'])
    source_size = reduce_product(a['source_shape']) if a['source_shape'] else input_size
    data = load_data_sets(os.path.join(data_root, data_name), source_size, is_image=data_is_image,
                          is_double=a['data_is_double'], raw_latent=raw_latent,
                          directed=a['directed'] or a['data_directed'], reorder=reorder, seed=seed)
    (data_normalizer, data_unnormalizer, ae_normalizer, ae_unnormalizer,
     latent_normalizer) = dist_normalizer(
        input_shape=a['input_shape'], ae_shape=a['ae_shape'], data_scale=data_scale,
        data_mean=data_mean, data_norm=a['data_norm'], latent_norm=latent_norm, data_type=a['data_type'])
    batch_size = a['batch_size']
    train_tr, val_tr = dist_transformer(a['source_shape'], a['input_shape'], data_random_crop, data_mirror)
    model, _ = construct_model(
        train_data_transformer=train_tr, val_data_transformer=val_tr,
        ae_transformer=dist_ae_transformer(a['input_shape'], a['ae_shape']),
        is_double=a.pop('data_is_double'), disable_double=a.pop('data_disable_double'), data=data,
        data_normalizer=data_normalizer, data_unnormalizer=data_unnormalizer,
        ae_normalizer=ae_normalizer, ae_unnormalizer=ae_unnormalizer,
        latent_normalizer=latent_normalizer, enable_input_producer=False, seed=seed, **a)
    checkpoint_dir = os.path.join(checkpoint_root, data_name, model.get_name())
    sample_dir = os.path.join(sample_root, data_name, model.get_name(), sample_type)
    _, start = load_model(model, checkpoint_dir)
    if start == 0:
        raise FileNotFoundError('no checkpoint under %s' % checkpoint_dir)
    fn = {'project': dist_sample, 'near': dist_sample_near, 'project_disc': dist_sample_project_disc}[sample_type]
    fn(sess=None, model=model, data=data.test, batch_size=batch_size, sample_dir=sample_dir, output_name='test')


def parse_args(argv=None):
    parser = dist_parser(batch_size=50)
    parser.add_argument('--sample-root', default='samples')
    parser.add_argument('--reorder', action='store_true')
    parser.add_argument('--sample-type', choices=('project', 'near', 'project_disc'), default='project')
    args = parser.parse_args(argv)
    dist_check_args(args)
    return args


def start(argv=None):
    logging.basicConfig(format='%(asctime)s [%(levelname)-5.5s] [%(name)s]  %(message)s', level=logging.WARNING)
    main(**vars(parse_args(argv)))


if __name__ == '__main__':
    start()
