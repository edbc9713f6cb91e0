"""A small build-owned SIFT-style extractor (numpy/scipy), used ONLY to turn the reference's two
sample photographs (img01.JPG / img02.JPG, named by BASELINE config C1) into keypoint +
descriptor fixtures: tests/golden/img01_img02_sift.npz.

The reference extracts SURF with OpenCV's nonfree module (main.cpp:22-40), which is out of scope
and unavailable; BASELINE config C1 names SIFT-128.  This extractor follows Lowe's scheme
(Gaussian scale space, DoG extrema, contrast + edge rejection, dominant orientation, 4x4x8
gradient histogram, normalise -> clip 0.2 -> renormalise -> x512 -> saturate to u8) closely
enough to give realistic, u8-valued 128-D descriptors; it makes no claim of bit-compatibility
with OpenCV's SIFT.  Not product code and not part of the timed path.
"""
import numpy as np
from scipy.ndimage import gaussian_filter, maximum_filter, minimum_filter


def _scale_space(img, n_oct, s=3, sigma0=1.6):
    k = 2.0 ** (1.0 / s)
    octaves = []
    base = gaussian_filter(img, np.sqrt(max(sigma0 ** 2 - 0.5 ** 2, 0.01)))
    for o in range(n_oct):
        g = [base]
        for i in range(1, s + 3):
            sig_prev = sigma0 * k ** (i - 1)
            sig_tot = sig_prev * k
            g.append(gaussian_filter(g[-1], np.sqrt(sig_tot ** 2 - sig_prev ** 2)))
        octaves.append(np.stack(g))
        base = g[s][::2, ::2]
    return octaves


def detect_and_describe(img, max_kp=4000, contrast=0.03, edge_r=10.0, s=3, sigma0=1.6):
    """img: 2-D float32 in [0,1].  Returns (kp_xy float32 [n,2], desc uint8 [n,128])."""
    n_oct = max(1, int(np.log2(min(img.shape))) - 4)
    octaves = _scale_space(img.astype(np.float32), n_oct, s, sigma0)
    kps = []
    for o, g in enumerate(octaves):
        dog = g[1:] - g[:-1]
        mx = maximum_filter(dog, size=(3, 3, 3))
        mn = minimum_filter(dog, size=(3, 3, 3))
        ext = ((dog == mx) | (dog == mn)) & (np.abs(dog) > contrast / s)
        ext[0] = ext[-1] = False
        ext[:, :8] = ext[:, -8:] = False
        ext[:, :, :8] = ext[:, :, -8:] = False
        for (i, y, x) in zip(*np.nonzero(ext)):
            d = dog[i]
            dxx = d[y, x + 1] + d[y, x - 1] - 2 * d[y, x]
            dyy = d[y + 1, x] + d[y - 1, x] - 2 * d[y, x]
            dxy = (d[y + 1, x + 1] - d[y + 1, x - 1] - d[y - 1, x + 1] + d[y - 1, x - 1]) / 4.0
            tr, det = dxx + dyy, dxx * dyy - dxy * dxy
            if det <= 0 or tr * tr * edge_r >= (edge_r + 1) ** 2 * det:
                continue
            kps.append((o, i, y, x, abs(d[y, x])))
    kps.sort(key=lambda t: -t[4])
    kps = kps[:max_kp]

    out_xy, out_desc = [], []
    k = 2.0 ** (1.0 / s)
    for (o, i, y, x, _) in kps:
        L = octaves[o][i]
        sig = sigma0 * k ** i
        # dominant orientation
        rad = int(round(3 * 1.5 * sig))
        if y - rad < 1 or x - rad < 1 or y + rad >= L.shape[0] - 1 or x + rad >= L.shape[1] - 1:
            continue
        win = L[y - rad - 1:y + rad + 2, x - rad - 1:x + rad + 2]
        gx = (win[1:-1, 2:] - win[1:-1, :-2]) * 0.5
        gy = (win[2:, 1:-1] - win[:-2, 1:-1]) * 0.5
        mag = np.sqrt(gx * gx + gy * gy)
        ang = np.arctan2(gy, gx)
        yy, xx = np.mgrid[-rad:rad + 1, -rad:rad + 1]
        wgt = np.exp(-(xx * xx + yy * yy) / (2 * (1.5 * sig) ** 2)) * mag
        hist = np.bincount(((ang + np.pi) / (2 * np.pi) * 36).astype(int).ravel() % 36, wgt.ravel(), 36)
        hist = (np.roll(hist, 1) + hist + np.roll(hist, -1)) / 3.0
        theta = (np.argmax(hist) + 0.5) / 36 * 2 * np.pi - np.pi
        # 4x4x8 descriptor on a rotated 16x16 grid (spacing 0.75*sig... window 3*sig per cell)
        cell = 3.0 * sig
        r2 = int(np.ceil(cell * 2.5 * np.sqrt(2))) + 1
        if y - r2 < 1 or x - r2 < 1 or y + r2 >= L.shape[0] - 1 or x + r2 >= L.shape[1] - 1:
            continue
        win = L[y - r2 - 1:y + r2 + 2, x - r2 - 1:x + r2 + 2]
        gx = (win[1:-1, 2:] - win[1:-1, :-2]) * 0.5
        gy = (win[2:, 1:-1] - win[:-2, 1:-1]) * 0.5
        yy, xx = np.mgrid[-r2:r2 + 1, -r2:r2 + 1].astype(np.float32)
        c, sn = np.cos(theta), np.sin(theta)
        u = (c * xx + sn * yy) / cell + 1.5      # cell coordinates in [-0.5, 3.5]
        v = (-sn * xx + c * yy) / cell + 1.5
        mag = np.sqrt(gx * gx + gy * gy) * np.exp(-((u - 1.5) ** 2 + (v - 1.5) ** 2) / (2 * 2.0 ** 2))
        ob = ((np.arctan2(gy, gx) - theta) % (2 * np.pi)) / (2 * np.pi) * 8
        ok = (u > -1) & (u < 4) & (v > -1) & (v < 4)
        u, v, ob, mag = u[ok], v[ok], ob[ok], mag[ok]
        u0, v0, o0 = np.floor(u).astype(int), np.floor(v).astype(int), np.floor(ob).astype(int)
        du, dv, do = u - u0, v - v0, ob - o0
        desc = np.zeros((4, 4, 8))
        for a, wa in ((0, 1 - dv), (1, dv)):
            for b, wb in ((0, 1 - du), (1, du)):
                for e, we in ((0, 1 - do), (1, do)):
                    vi, ui, oi = v0 + a, u0 + b, (o0 + e) % 8
                    m = (vi >= 0) & (vi < 4) & (ui >= 0) & (ui < 4)
                    np.add.at(desc, (vi[m], ui[m], oi[m]), (mag * wa * wb * we)[m])
        d = desc.ravel()
        n = np.linalg.norm(d)
        if n < 1e-9:
            continue
        d = np.minimum(d / n, 0.2)
        d = d / np.linalg.norm(d)
        out_desc.append(np.clip(np.rint(d * 512), 0, 255).astype(np.uint8))
        scale = 2.0 ** o
        out_xy.append((x * scale, y * scale))
    return np.array(out_xy, np.float32).reshape(-1, 2), np.array(out_desc, np.uint8).reshape(-1, 128)
