"""The scalar functions the model descriptors carry as default hyper-priors
(include/microscopes/common/scalar_functions.hpp:34-74, microscopes/common/scalar_functions.pyx): callables with an
`input_dim()`, float arithmetic as upstream, picklable the way the reference's are (`__reduce__`).  Host-side and tiny:
they are evaluated a handful of times per hyper-parameter move, never per row."""
import numpy as np

_f = np.float32


class scalar_function(object):
    _dim = 1

    def input_dim(self):
        return self._dim


class log_exponential(scalar_function):
    """log of the Exponential(lam) density; -inf for x < 0 (scalar_functions.hpp:34-46)"""

    def __init__(self, lam):
        self._lam = float(_f(lam))
        self._log_lam = _f(np.log(_f(lam)))

    def __call__(self, x):
        x = _f(x)
        if x < 0:
            return float("-inf")
        return float(self._log_lam - _f(self._lam) * x)

    def __reduce__(self):
        return (_reconstruct_log_exponential, (self._lam,))


class log_normal(scalar_function):
    """log of the Normal(mu, sigma2) density, sigma2 the variance (scalar_functions.hpp:48-61)"""

    def __init__(self, mu, sigma2):
        if not sigma2 > 0.0:
            raise ValueError("sigma2 cannot be zero")
        self._mu, self._sigma2 = float(_f(mu)), float(_f(sigma2))
        self._lgc = _f(-0.5) * _f(np.log(_f(2.0 * np.pi) * _f(sigma2)))
        self._half_inv = _f(0.5) / _f(sigma2)

    def __call__(self, x):
        d = _f(x) - _f(self._mu)
        return float(self._lgc - self._half_inv * d * d)

    def __reduce__(self):
        return (_reconstruct_log_normal, (self._mu, self._sigma2))


class _log_noninformative_beta_prior(scalar_function):
    """-2.5 log(alpha + beta), a proper non-informative prior for the beta distribution; -inf outside the positive
    quadrant (scalar_functions.hpp:63-74)"""
    _dim = 2

    def __call__(self, alpha, beta):
        if alpha <= 0.0 or beta <= 0.0:
            return float("-inf")
        return float(_f(-2.5) * _f(np.log(_f(alpha) + _f(beta))))

    def __reduce__(self):
        return (_reconstruct_log_noninformative_beta_prior, ())


def _reconstruct_log_exponential(lam):
    return log_exponential(lam)


def _reconstruct_log_normal(mu, sigma2):
    return log_normal(mu, sigma2)


def _reconstruct_log_noninformative_beta_prior():
    return log_noninformative_beta_prior


log_noninformative_beta_prior = _log_noninformative_beta_prior()
