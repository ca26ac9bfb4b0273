"""The Monomer-data model: ``Dist`` (cfl/models/dist.py:92-327) and its
``construct_model`` (cfl/models/dist.py:330-459), on the fused HIP pair path.

FCEncoder (cfl/models/dist.py:12-68) = two plain fully-connected heads with biases,
``latent_outputs`` [D, L] and ``pcd_outputs`` [D, L*K]; distance = PCD soft-min
(cfl/models/dist.py:70-89); loss = reg + BCE(pos, 1) + BCE(neg, 0) through the
learned threshold, always (cfl/models/dist.py:253-269); one Adam over encoder +
threshold (cfl/models/dist.py:286-293).
"""
from argparse import Namespace

from .. import hipabi as H
from ..utils import reduce_product
from .base import PairModel


class Dist(PairModel):
    MODEL_SCOPE = 'Dist'
    ENCODER_SCOPES = ('Encoder',)
    HEAD_SCOPES = {
        'outputs': ('latent_outputs/fully_connected', {'W': 'weights', 'b': 'biases'}),
        'proto': ('pcd_outputs/fully_connected', {'W': 'weights', 'b': 'biases'}),
    }

    def __init__(self, input_shape, latent_size, num_components, batch_size, lr, beta1, beta2,
                 batches=None, val_batches=None, normalize_value=None, data_normalizer=None,
                 data_unnormalizer=None, reg_const=0.0, name='Dist', run_tag=None, reuse=False,
                 seed=0, device=None):
        self.is_double = False
        self.input_shape = tuple(input_shape)
        self.latent_size = latent_size
        self.reg_const = reg_const
        self.batch_size = batch_size
        self.num_components = num_components
        self.normalize_value = normalize_value
        self.run_tag = run_tag
        self.lr, self.beta1, self.beta2 = lr, beta1, beta2
        self.data_normalizer = data_normalizer
        self.data_unnormalizer = data_unnormalizer
        self.ae_shape = self.input_shape
        norm = data_normalizer.to_cfl_norm() if data_normalizer is not None else H.make_norm()
        self._setup_engine(
            reduce_product(self.input_shape), latent_size, num_components, 'pcd',
            weight_norm=False, has_bias=True, act_type=None, directed=False, norm=norm,
            loss=H.make_loss(use_threshold=True, reg_const=reg_const), lr=lr, beta1=beta1,
            beta2=beta2, batch_size=batch_size, seed=seed, device=device)

    def get_name(self):
        """cfl/models/dist.py:192-199, e.g. linear_dist_ls_10_nc_4_reg_0.0_norm_58.388599."""
        name = 'linear_dist_ls_{}_nc_{}_reg_{}_norm_{}'.format(
            self.latent_size, self.num_components, self.reg_const, self.normalize_value)
        if self.run_tag:
            name += '_run_' + self.run_tag
        return name


def construct_model(input_shape, latent_size, num_components, lr, beta1, beta2, batch_size,
                    normalize_value, reg_const=0.0, data_normalizer=None, data_unnormalizer=None,
                    data=None, run_tag=None, seed=0, device=None):
    """(model, aux).  ``aux`` replaces the reference's FIFO queues + enqueue ops with
    the two batch sources the train loop draws from: aux.train / aux.val."""
    model = Dist(input_shape=input_shape, latent_size=latent_size, num_components=num_components,
                 reg_const=reg_const, batch_size=batch_size, lr=lr, beta1=beta1, beta2=beta2,
                 normalize_value=normalize_value, data_normalizer=data_normalizer,
                 data_unnormalizer=data_unnormalizer, run_tag=run_tag, seed=seed, device=device)
    aux = None if data is None else Namespace(train=data.train, val=data.val)
    return model, aux
