"""CPU stand-in for `pivp_amd.Model` with the same training protocol: a TEST DOUBLE (tests/test_parallel_gloo.py, `bench.py --dry`),
not part of the product package."""
import torch


def local_gradient(seed, g, n):
    """the pseudo-random gradient of group g on the rank whose stub was built with `random_seed=seed` (tests rebuild every rank's)"""
    gen = torch.Generator().manual_seed(1000 * int(seed) + g)
    return torch.randn(n, generator=gen) * (1.0 + g)


class HostStubModel(object):
    """CPU stand-in for `Model` with the same training protocol (`_ensure_grads`, `grad_group_ranges`, `cleargrads`,
    `backward(on_group)`): the gradient of group g is `value * (g + 1)` everywhere (or pseudo-random: `random_seed`).  It lets the data-parallel host logic run
    under gloo with no GPU: tests/test_parallel_gloo.py and `bench.py --dry`.  `fail_in_group` makes the callback of that group
    raise, as a failing rank would; `skip_groups` leaves groups unannounced."""

    def __init__(self, sizes=(1000, 300, 70, 5000, 64, 1), value=1.0, fail_in_group=None, skip_groups=(), precision='fp32', random_seed=None):
        self.precision = precision
        self.sizes = list(sizes)
        self.value = float(value)
        self.fail_in_group = fail_in_group
        self.skip_groups = set(skip_groups)
        self._flat_params = torch.zeros(sum(self.sizes))
        self._flat = torch.zeros(sum(self.sizes))
        self.announced = []
        self.random_seed = random_seed      # not None: group g's gradient is `local_gradient(random_seed, g, n)` instead of a constant

    def _ensure_grads(self):
        return self._flat

    def cleargrads(self):
        self._flat.zero_()

    def grad_group_ranges(self):
        out, o = [], 0
        for n in self.sizes:
            out.append((o, o + n))
            o += n
        return out

    def backward(self, on_group=None):
        errors = []
        for g, (a, b) in enumerate(self.grad_group_ranges()):
            self._flat[a:b] += self.value * (g + 1) if self.random_seed is None else local_gradient(self.random_seed, g, b - a)
            if on_group is None or g in self.skip_groups:
                continue
            self.announced.append(g)
            try:                                     # like Model.backward: the sweep continues, the first error is raised afterwards
                if g == self.fail_in_group:
                    raise RuntimeError('injected failure in group %d' % g)
                on_group(g)
            except BaseException as e:               # noqa: B902
                errors.append(e)
        if errors:
            raise errors[0]
