"""TEST INFRASTRUCTURE ONLY (oracle).

Twisted-Edwards addition split into three degree-2 layers, and the AlgFn
combinators, restated from
  /root/reference/src/cleanup/utils/twisted_edwards_ops.rs:10-80  (fns)
  /root/reference/src/cleanup/utils/twisted_edwards_ops.rs:150-156 (deg, n_ins, n_outs)
  /root/reference/src/cleanup/utils/algfn.rs:129-292 (Id / Repeated / Stacked / BitCheck)
Values are canonical ints mod P.
"""
from .field import P, mul_by_a, mul_by_d


def affine_twisted_edwards_add_l1(a):
    x1, y1, x2, y2 = a
    return [x1 * y2 % P, x2 * y1 % P, (y1 * y2 - mul_by_a(x1 * x2 % P)) % P]


def affine_twisted_edwards_add_l2(a):
    x1y2, x2y1, yy = a
    return [(x1y2 + x2y1) % P, yy, x1y2 * x2y1 % P]


def affine_twisted_edwards_add_l3(a):
    x, y, xy = a
    dxy = mul_by_d(xy)
    m = (1 - dxy) % P
    p = (1 + dxy) % P
    return [m * x % P, p * y % P, m * p % P]


def twisted_edwards_add_l1(a):
    x1, y1, z1, x2, y2, z2 = a
    return [x1 * y2 % P, x2 * y1 % P, (y1 * y2 - mul_by_a(x1 * x2 % P)) % P, z1 * z2 % P]


def twisted_edwards_add_l2(a):
    x1y2, x2y1, yy, zz = a
    return [(x1y2 + x2y1) * zz % P, yy * zz % P, zz * zz % P, x1y2 * x2y1 % P]


def twisted_edwards_add_l3(a):
    x, y, z2, xy = a
    dxy = mul_by_d(xy)
    m = (z2 - dxy) % P
    p = (z2 + dxy) % P
    return [m * x % P, p * y % P, m * p % P]


def triangle_twisted_edwards_add_l1(pts):
    assert len(pts) == 12
    a, b, c, d = pts[0:3], pts[3:6], pts[6:9], pts[9:12]
    return (twisted_edwards_add_l1(a + c) + twisted_edwards_add_l1(b + d)
            + twisted_edwards_add_l1(c + d))


class AlgFn:
    def __init__(self, name, deg, n_ins, n_outs, fn):
        self.name, self.deg, self.n_ins, self.n_outs, self.fn = name, deg, n_ins, n_outs, fn

    def exec(self, args):
        return self.fn(list(args[: self.n_ins]))


AFF_L1 = AlgFn("affine_twisted_edwards_add_l1", 2, 4, 3, affine_twisted_edwards_add_l1)
AFF_L2 = AlgFn("affine_twisted_edwards_add_l2", 2, 3, 3, affine_twisted_edwards_add_l2)
AFF_L3 = AlgFn("affine_twisted_edwards_add_l3", 2, 3, 3, affine_twisted_edwards_add_l3)
PROJ_L1 = AlgFn("twisted_edwards_add_l1", 2, 6, 4, twisted_edwards_add_l1)
PROJ_L2 = AlgFn("twisted_edwards_add_l2", 2, 4, 4, twisted_edwards_add_l2)
PROJ_L3 = AlgFn("twisted_edwards_add_l3", 2, 4, 3, twisted_edwards_add_l3)
TRI_L1 = AlgFn("triangle_twisted_edwards_add_l1", 2, 12, 12, triangle_twisted_edwards_add_l1)


def IdAlgFn(n):
    return AlgFn("id%d" % n, 1, n, n, lambda a: list(a))


def BitCheckFn():
    return AlgFn("bitcheck", 2, 1, 1, lambda a: [(a[0] * a[0] - a[0]) % P])


def RepeatedAlgFn(f, count):
    def fn(a):
        out = []
        for i in range(count):
            out += f.exec(a[i * f.n_ins:(i + 1) * f.n_ins])
        return out
    return AlgFn("rep%d[%s]" % (count, f.name), f.deg, f.n_ins * count, f.n_outs * count, fn)


def StackedAlgFn(f1, f2):
    def fn(a):
        return f1.exec(a[: f1.n_ins]) + f2.exec(a[f1.n_ins: f1.n_ins + f2.n_ins])
    return AlgFn("stack[%s|%s]" % (f1.name, f2.name), max(f1.deg, f2.deg),
                 f1.n_ins + f2.n_ins, f1.n_outs + f2.n_outs, fn)
