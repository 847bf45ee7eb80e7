"""The solver-free parts of the policy mirror (deqmpc/policies.py:532-564,719-848) -- shapes and losses on the CPU."""
import argparse
import types

import pytest
import torch

from diff_qp_mpc_amd import policies


def _args(out_type, T=5, nq=1):
    return argparse.Namespace(T=T, nq=nq, hdim=16, policy_out_type=out_type, device="cpu", deq=False, en_qp_solve=False)


@pytest.mark.parametrize("out_type", [0, 1, 2, 3])
def test_nnpolicy_shapes_and_bc_loss(out_type):
    env = types.SimpleNamespace(nx=2, nu=1, dt=0.05)
    torch.manual_seed(0)
    pol = policies.NNPolicy(_args(out_type), env)
    x = torch.randn(7, 2)
    states, actions = pol(x)
    assert (states is None) == (out_type == 0) and (actions is None) == (out_type in (1, 3))
    if states is not None:
        assert states.shape == (7, 5, 2)
    if actions is not None:
        assert actions.shape == (7, 5, 1)
    if out_type == 3:           # velocities are the forward differences of the predicted configurations
        torch.testing.assert_close(states[:, :-1, 1], (states[:, 1:, 0] - states[:, :-1, 0]) / 0.05)
    gs, ga, mask = torch.randn(7, 5, 2), torch.randn(7, 5, 1), torch.ones(7, 5)
    mask[:, -1] = 0
    loss, end = policies.compute_loss(pol, gs, ga, mask, (states, actions), _args(out_type))
    want = 0.0
    if out_type in (0, 2):
        want = want + ((actions - ga).abs() * mask[:, :, None]).sum(-1).mean()
    if out_type in (1, 2):
        want = want + ((states - gs).abs() * mask[:, :, None]).sum(-1).mean()
    if out_type == 3:
        want = want + ((states[..., :1] - gs[..., :1]).abs() * mask[:, :, None]).sum(-1).mean()
    torch.testing.assert_close(loss, want)
    assert float(end) == 0.0


def test_ffdnetwork_offsets_from_current_configuration():
    env = types.SimpleNamespace(nx=4, nu=1, dt=0.05)
    net = policies.FFDNetwork(_args(1, T=6, nq=2), env)
    with torch.no_grad():
        net.fc3.weight.zero_(); net.fc3.bias.zero_()
    x = torch.randn(3, 4)
    out = net(x)
    assert out.shape == (3, 6, 2)
    torch.testing.assert_close(out, x[:, None, :2].expand(3, 6, 2))
    assert set(k.split(".")[0] for k in net.state_dict()) >= {"fc1", "ln1", "fc2", "ln2", "fc3", "net"}


def test_deq_losses():
    pol = types.SimpleNamespace(nq=1, out_type=2, model=types.SimpleNamespace(nq=1))
    gs, ga, mask = torch.randn(4, 3, 2), torch.randn(4, 3, 1), torch.ones(4, 3)
    trajs = [(None, torch.randn(4, 3, 2), torch.randn(4, 3, 1)) for _ in range(3)]
    a = argparse.Namespace(deq=True, en_qp_solve=True)
    loss, end = policies.compute_loss(pol, gs, ga, mask, trajs, a)
    per = [((ns - gs).abs().sum(-1).mean() + (na - ga).abs().sum(-1).mean()) for _, ns, na in trajs]
    torch.testing.assert_close(loss, sum(per)); torch.testing.assert_close(end, per[-1])
    a.en_qp_solve = False                                # pre-training: states only, through policy.model
    loss, end = policies.compute_loss(pol, gs, ga, mask, trajs, a)
    per = [(ns - gs).abs().sum(-1).mean() for _, ns, _ in trajs]
    torch.testing.assert_close(loss, sum(per)); torch.testing.assert_close(end, per[-1])


def test_non_mlp_layers_raise_with_the_reason():
    env = types.SimpleNamespace(nx=2, nu=1, dt=0.05)
    a = argparse.Namespace(T=5, nq=1, hdim=16, layer_type="gcn", deq_out_type=1)
    with pytest.raises(NotImplementedError, match="gcn"):
        policies.DEQLayer(a, env)


def test_sliced_linear_matches_linear_on_cpu():
    """policies.SlicedLinear (the DEQLayer linears' weight gradient as a batched product over row slices): same forward and
    the same three gradients as torch.nn.functional.linear, here in float64 on the CPU where the sums agree to round-off."""
    torch.manual_seed(0)
    for rows in (1024, 1536, 1000):                 # 2 slices of 512, 1 slice (1536 = 3 x 512 has no even split), odd
        x = torch.randn(rows, 7, dtype=torch.float64, requires_grad=True)
        w = torch.randn(5, 7, dtype=torch.float64, requires_grad=True)
        b = torch.randn(5, dtype=torch.float64, requires_grad=True)
        g = torch.randn(rows, 5, dtype=torch.float64)
        y0 = torch.nn.functional.linear(x, w, b)
        want = torch.autograd.grad(y0, (x, w, b), g)
        y1 = policies.SlicedLinear.apply(x, w, b)
        got = torch.autograd.grad(y1, (x, w, b), g)
        torch.testing.assert_close(y1, y0, rtol=0, atol=0)
        for a, c in zip(got, want):
            torch.testing.assert_close(a, c, rtol=1e-12, atol=1e-12)
        y2 = policies.SlicedLinear.apply(x, w, None)        # no bias
        torch.testing.assert_close(torch.autograd.grad(y2, w, g)[0], want[1], rtol=1e-12, atol=1e-12)
