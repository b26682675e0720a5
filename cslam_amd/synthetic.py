"""The synthetic inputs of BASELINE.md section 3, bit for bit, for bench.py and the full-size tests:

    bank     numpy.random.default_rng(1234 + robot_id).standard_normal((N, D)).astype(float32), rows L2-normalised in float32
    queries  the same from default_rng(4321 + robot_id)
    frames   uniform uint8 [480, 640, 3] from default_rng(7 + i), one generator per frame i

Generated on the host in row blocks (a Generator continues its stream from call to call, so the blocks concatenate to what
one call would return) and uploaded once; CPU checkers and the GPU path therefore see identical bits.  Nothing here is on
the product path."""
import numpy as np


def unit_rows(seed, n, dim, block=8192):
    """[n, dim] float32, rows of default_rng(seed).standard_normal(...).astype(float32) normalised in float32."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, dim), dtype=np.float32)
    for r0 in range(0, n, block):
        r1 = min(n, r0 + block)
        x = rng.standard_normal((r1 - r0, dim)).astype(np.float32)
        x /= np.linalg.norm(x, axis=1, keepdims=True)
        out[r0:r1] = x
    return out


def bank(robot_id, n, dim):
    return unit_rows(1234 + robot_id, n, dim)


def queries(robot_id, n, dim):
    return unit_rows(4321 + robot_id, n, dim)


def frames(first, count, height=480, width=640):
    """[count, height, width, 3] uint8: frame i = default_rng(7 + i).integers(0, 256, (height, width, 3), dtype=uint8)."""
    out = np.empty((count, height, width, 3), dtype=np.uint8)
    for j in range(count):
        out[j] = np.random.default_rng(7 + first + j).integers(0, 256, (height, width, 3), dtype=np.uint8)
    return out
