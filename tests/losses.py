"""Callable losses for the `loss=` tests (least_squares' contract: z = (f / f_scale)^2 -> array (3, m) of rho, rho', rho'';
scipy/optimize/_lsq/least_squares.py:160-227).  Shared by tests/golden/make_golden.py --callable (which runs the reference with them) and the tests."""
import numpy as np


def charbonnier_quarter(z):
    """Generalised Charbonnier loss with exponent 1/4 -- rho(z) = 4 ((1 + z)^(1/4) - 1): between soft_l1 (exponent 1/2) and cauchy (-> 0);
    not one of scipy's five names."""
    t = 1.0 + np.asarray(z, dtype=float)
    rho = np.empty((3,) + t.shape)
    rho[0] = 4.0 * (t**0.25 - 1.0)
    rho[1] = t**-0.75
    rho[2] = -0.75 * t**-1.75
    return rho


def soft_l1_as_callable(z):
    """scipy's soft_l1 (least_squares.py: soft_l1) written as a callable: the callable path must reproduce the built-in one."""
    t = 1.0 + np.asarray(z, dtype=float)
    rho = np.empty((3,) + t.shape)
    rho[0] = 2.0 * (t**0.5 - 1.0)
    rho[1] = t**-0.5
    rho[2] = -0.5 * t**-1.5
    return rho
