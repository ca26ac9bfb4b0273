"""python -m cfl.bin.predict_dist -- score every labeled pair of train / val / test
with the best_acc_model checkpoint of cfl.bin.train_dist and write the
'<id1> match <id2> <score>' files that cfl.bin.evaluate_total reads
(drop-in for cfl/bin/predict_dist.py:17-82)."""
import logging
import os

from ..input_data import load_data_sets
from ..models.dist import construct_model
from ..ops import normalizer, unnormalizer
from ..utils import dist_predict, load_model, monomer_parser, reduce_product

logger = logging.getLogger(__name__)


def predict_monomer(data_name, data_root, checkpoint_root, log_root, predict_root, run_tag, seed,
                    normalize_value, input_shape, batch_size, num_components, latent_size, lr, beta1,
                    beta2, reg_const):
    input_shape = tuple(input_shape)
    data = load_data_sets(os.path.join(data_root, data_name), reduce_product(input_shape), seed=seed)
    model, _ = construct_model(
        input_shape=input_shape, latent_size=latent_size, normalize_value=normalize_value, lr=lr,
        beta1=beta1, beta2=beta2, num_components=num_components, batch_size=batch_size, data=data,
        run_tag=run_tag, reg_const=reg_const, data_normalizer=normalizer(normalize_value, 0., None, None),
        data_unnormalizer=unnormalizer(normalize_value, 0.), seed=seed)
    checkpoint_dir = os.path.join(checkpoint_root, data_name, model.get_name())
    predict_dir = os.path.join(predict_root, data_name, model.get_name())
    load_model(model, os.path.join(checkpoint_dir, 'best_acc_model'))
    for split, name in (('train', 'predict_train_acc.txt'), ('val', 'predict_val_acc.txt'),
                        ('test', 'predict_acc.txt')):
        dist_predict(None, model, getattr(data, split), batch_size, predict_dir, name)


def parse_args(argv=None):
    parser = monomer_parser()
    parser.add_argument('--predict-root', default='predicts')
    return parser.parse_args(argv)


def main(argv=None):
    logging.basicConfig(format='%(asctime)s [%(levelname)-5.5s] [%(name)s]  %(message)s',
                        level=logging.WARNING)
    predict_monomer(**vars(parse_args(argv)))


if __name__ == '__main__':
    main()
