"""
Batched collocation (basis) matrix: the front half of the reference's ``least_squares``
(bspy/_spline_fitting.py:736-751) and ``solve_ode`` collocation - SURVEY.md 8f-3.

The reference fills ``A[iRow, ix - order : ix] = bspline_values(...)`` one row at a time in
Python; a repeated parameter value means "next derivative" (Hermite conditions).  Here the
basis rows of all parameters come from one batched GPU call per derivative order
(``bsk_bspline_values``); the host only places them into the matrix.
"""
import numpy as np

from .device_spline import bspline_values_batch


def derivative_orders(uValues):
    """Derivative order of every row: the number of immediately preceding equal parameter
    values (reference _spline_fitting.py:741-748)."""
    u = np.asarray(uValues)
    orders = np.zeros(len(u), np.int64)
    for i in range(1, len(u)):
        if u[i] == u[i - 1]:
            orders[i] = orders[i - 1] + 1
    return orders


def collocation_matrix(knots, order, uValues, dense=True):
    """Matrix A with ``A[i, ix_i - order : ix_i] = basis row of uValues[i]`` (derivative rows for
    repeated values), shape (len(uValues), len(knots) - order).

    dense=False returns the banded form ``(ix, rows)``: first column index + ``order`` values
    per row, which is what the matrix holds."""
    knots = np.asarray(knots)
    dt = np.float32 if knots.dtype == np.float32 else np.float64
    u = np.ascontiguousarray(uValues, dt).ravel()
    order = int(order)
    ncols = len(knots) - order
    derivs = derivative_orders(u)
    ix = np.empty(len(u), np.int32)
    rows = np.empty((len(u), order), dt)
    for dv in np.unique(derivs):
        sel = np.flatnonzero(derivs == dv)
        ix[sel], rows[sel] = bspline_values_batch(knots, order, u[sel], int(dv))
    if not dense:
        return ix - order, rows
    A = np.zeros((len(u), ncols), dt)
    cols = (ix - order)[:, None] + np.arange(order)[None, :]
    A[np.arange(len(u))[:, None], cols] = rows
    return A
