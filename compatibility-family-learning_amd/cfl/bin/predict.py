"""python -m cfl.bin.predict -- score files for cfl.bin.evaluate_total from the
best_model (AUC-selected) and best_acc_model checkpoints of cfl.bin.train
(drop-in for cfl/bin/predict.py:19-199)."""
import logging
import os

from ..input_data import load_data_sets
from ..models.cfl import construct_model
from ..ops import dist_ae_transformer, dist_normalizer, dist_transformer
from ..utils import dist_check_args, dist_parser, dist_predict, load_model, reduce_product

logger = logging.getLogger(__name__)


def main(predict_root, data_name, data_root, checkpoint_root, log_root, seed, data_mirror,
         data_random_crop, data_is_image, raw_latent, data_scale, data_mean, latent_norm,
         **model_args):
    a = model_args
    input_size = reduce_product(a['input_shape'])
    source_size = reduce_product(a['source_shape']) if a['source_shape'] else input_size
    data = load_data_sets(os.path.join(data_root, data_name), source_size, is_image=data_is_image,
                          is_double=a['data_is_double'], raw_latent=raw_latent,
                          directed=a['directed'] or a['data_directed'], seed=seed)
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
    predict_dir = os.path.join(predict_root, data_name, model.get_name())
    logger.warning('run with %s', model.get_name())
    for sub, suffix in (('best_model', ''), ('best_acc_model', '_acc')):
        load_model(model, os.path.join(checkpoint_dir, sub))
        for split, stem in (('train', 'predict_train'), ('val', 'predict_val'), ('test', 'predict')):
            dist_predict(None, model, getattr(data, split), batch_size, predict_dir,
                         '{}{}.txt'.format(stem, suffix))


def parse_args(argv=None):
    parser = dist_parser(batch_size=500)
    parser.add_argument('--predict-root', default='predicts')
    args = parser.parse_args(argv)
    dist_check_args(args)
    return args


def start(argv=None):
    logging.basicConfig(format='%(asctime)s [%(levelname)-5.5s] [%(name)s]  %(message)s',
                        level=logging.WARNING)
    main(**vars(parse_args(argv)))


if __name__ == '__main__':
    start()
