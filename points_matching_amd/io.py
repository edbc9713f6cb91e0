"""`.pmm` matrix files read by the C++ host tool (host/pm_cli.cpp): magic "PMM1", int32 rows,
int32 cols, int32 dtype (0 = float32, 1 = uint8), row-major data."""
import numpy as np


def save_pmm(path, a):
    a = np.ascontiguousarray(a)
    assert a.ndim == 2 and a.dtype in (np.float32, np.uint8)
    with open(path, "wb") as f:
        f.write(b"PMM1")
        f.write(np.array([a.shape[0], a.shape[1], 0 if a.dtype == np.float32 else 1], "<i4").tobytes())
        f.write(a.tobytes())


def load_pmm(path):
    with open(path, "rb") as f:
        assert f.read(4) == b"PMM1"
        r, c, dt = np.frombuffer(f.read(12), "<i4")
        return np.frombuffer(f.read(), np.float32 if dt == 0 else np.uint8).reshape(r, c).copy()
