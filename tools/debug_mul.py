import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import forge_ec_amd as F
import vectors as V
from oracle import c_oracle as C
ctx = F.Context(0)
edges = V.edge_field_values(0)
a = np.array([V.limbs_of(x) for x in edges for _ in edges], dtype=np.uint64)
b = np.array([V.limbs_of(y) for _ in edges for y in edges], dtype=np.uint64)
got = ctx.field_op(0, 2, a, b)
for i in range(a.shape[0]):
    want = C.field_op(0, 'mul', a[i], b[i])
    if not np.array_equal(got[i], want):
        print("a=%x b=%x\n   got  %s\n   want %s" % (V.int_of(a[i]), V.int_of(b[i]), [hex(int(v)) for v in got[i]], [hex(int(v)) for v in want]))
