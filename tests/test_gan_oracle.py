"""CPU checks that pin the MrCGAN oracle (oracle/gan_oracle.py, oracle/conv_oracle.py) by
independent restatements: the reference's literal split/concat/reshape sub-pixel recipe, the
adjoint identity that defines conv2d_transpose, finite differences of the gradient penalty's
gradient (the double backward), and closed forms of the scalar losses."""
import numpy as np
import torch

from oracle import conv_oracle as CO
from oracle import gan_oracle as GO


def test_subpixel_equals_split_concat_reshape():
    """cfl/layers.py:240-246: split channels in `scale` groups, concat along W, reshape."""
    rng = np.random.RandomState(0)
    x = rng.randn(2, 3, 5, 8)
    r = 2
    parts = np.split(x, r, axis=3)
    y = np.concatenate(parts, axis=2).reshape(2, 3 * r, 5 * r, 8 // (r * r))
    got = CO.conv2d_subpixel(torch.tensor(x), r).numpy()
    assert np.array_equal(got, y)


def test_conv_transpose_is_adjoint_of_same_conv():
    rng = np.random.RandomState(1)
    V = torch.tensor(rng.randn(5, 5, 3, 4))       # [KH,KW,Co_t,Ci_t] == HWIO of the forward conv
    g = torch.tensor(1 + 0.1 * rng.randn(3))
    x = torch.tensor(rng.randn(2, 4, 4, 4))        # input of the transposed layer
    y = torch.tensor(rng.randn(2, 8, 8, 3))        # input of the forward conv
    t = GO.conv2d_transpose_weight_norm(x, V, g, None, 2)
    assert tuple(t.shape) == (2, 8, 8, 3)
    n2 = (V * V).sum(dim=(0, 1, 3), keepdim=True)
    W = V * torch.rsqrt(n2) * g.reshape(1, 1, -1, 1)
    f = CO.conv2d_weight_norm(y, W, None, None, 2)  # wn_filter renormalises; undo below
    # build the plain forward conv with filter W directly
    import torch.nn.functional as F
    _, pt, pb = CO.same_pads(8, 5, 2)
    f = F.conv2d(F.pad(y.permute(0, 3, 1, 2), (pt, pb, pt, pb)), W.permute(3, 2, 0, 1), stride=2).permute(0, 2, 3, 1)
    assert abs(float((f * x).sum() - (t * y).sum())) < 1e-9


def test_gradient_penalty_double_backward_matches_finite_differences():
    rng = np.random.RandomState(2)
    shape, Ld = (8, 8, 3), 4
    p = GO.init_sr_discriminator(shape, Ld, rng)
    p = {k: torch.tensor(v) for k, v in p.items()}
    x = torch.tensor(np.tanh(rng.randn(3, 192)))

    def gp(params):
        xh = x.clone().requires_grad_(True)
        o, _ = GO.sr_discriminator(xh, params, shape)
        g, = torch.autograd.grad(o.sum(), xh, create_graph=True)
        return 0.5 * ((torch.sqrt((g * g).sum(1)) - 1.0) ** 2).mean()

    name = 'conv1/Conv_1/V'
    q = {k: v.clone().requires_grad_(k == name) for k, v in p.items()}
    grad, = torch.autograd.grad(gp(q), q[name])
    idx = [(0, 1, 2, 3), (2, 2, 30, 7), (1, 0, 5, 20)]
    for i in idx:
        h = 1e-5
        qp = {k: v.clone() for k, v in p.items()}
        qm = {k: v.clone() for k, v in p.items()}
        qp[name][i] += h
        qm[name][i] -= h
        fd = (float(gp(qp)) - float(gp(qm))) / (2 * h)
        assert abs(fd - float(grad[i])) <= 1e-6 + 1e-4 * abs(fd), (i, fd, float(grad[i]))


def test_scalar_losses_closed_forms():
    x = torch.tensor([0.0, 2.0, -3.0], dtype=torch.float64)
    assert abs(float(GO.bce(x, 1.0)) - float(np.mean(np.log1p(np.exp(-np.array([0.0, 2.0, -3.0])))))) < 1e-12
    assert abs(float(GO.bce(x, 0.0)) - float(np.mean(np.log1p(np.exp(np.array([0.0, 2.0, -3.0])))))) < 1e-12
    a = GO.AdamTF({'w': torch.zeros(2)}, 2e-4, 0.5)
    params = {'w': torch.zeros(2, dtype=torch.float64)}
    a.apply(params, {'w': torch.tensor([1.0, -2.0], dtype=torch.float64)})
    # first Adam step moves every element by ~lr against the gradient sign
    assert np.allclose(params['w'].numpy(), [-2e-4, 2e-4], rtol=1e-4)


def test_generator_shapes():
    rng = np.random.RandomState(3)
    for gt, shape in (('srgan', (64, 64, 3)), ('conv', (28, 28, 1))):
        p = GO.GENERATORS[gt][0](shape, 84, rng)
        n = sum(int(np.prod(v.shape)) for v in p.values())
        assert n > 0
    p = GO.init_sr_generator((64, 64, 3), 84, rng)
    assert p['fc1/fully_connected/V'].shape == (84, 1024)
    assert p['subpixel_block1/Conv/V'].shape == (3, 3, 64, 2048)
    assert p['subpixel_block3/Conv/V'].shape == (3, 3, 256, 512)
    assert p['outputs/Conv/V'].shape == (3, 3, 128, 12)
    d = GO.init_sr_discriminator((64, 64, 3), 64, rng)
    assert d['conv4/Conv_4/V'].shape == (4, 4, 256, 512)
    assert d['disc_outputs/fully_connected/V'].shape == (2048, 1)
