"""the subset of the reference's tiGadgets.py that the Newton driver uses (tiGadgets.py:5-37,
67-70), operating on device vectors through the C ABI.  Arguments are
`femcy_amd.backend.DeviceVector` handles of one context."""


def _same_ctx(*vs):
    ctx = vs[0].ctx
    assert all(v.ctx is ctx for v in vs), "vectors belong to different contexts"
    return ctx


def c_equals_a_minus_b(c, a, b):
    """c = a - b"""
    _same_ctx(c, a, b).vec_sub(c.id, a.id, b.id)


def a_equals_b_plus_c_mul_d(a, b, c: float, d):
    """a = b + c * d"""
    _same_ctx(a, b, d).vec_axpy(a.id, b.id, c, d.id)


def field_abs_max(f) -> float:
    return f.ctx.vec_absmax(f.id)


def field_norm(f) -> float:
    """sqrt(sum f^2 / N): the reference's "modified 2nd norm"."""
    return f.ctx.vec_norm(f.id)


def field_multiply(field, num: float):
    field.ctx.vec_scale(field.id, num)
