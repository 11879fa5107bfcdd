"""NumPy model of the index algebra the HIP kernels use (helicon_amd/csrc/helicon_hip.hip).

Not a product path and not the oracle: it exists so the kernel DESIGN (Stockham stage indexing
with 8 points per lane, two-real-columns-per-complex-FFT separation, the packed ky=0/ky=N/2
row, Hermitian half-plane weights, Pearson from weighted moments) can be checked on the CPU
against np.fft / the oracle before a GPU minute is spent.  Each function mirrors one device
function of the .hip file, vectorised over the lane index ``t``.
"""
import numpy as np

RADICES = {32: (8, 4), 64: (8, 8), 128: (8, 8, 2), 256: (8, 8, 4), 512: (8, 8, 8), 1024: (8, 8, 8, 2)}


def fft_lanes(v):
    """v[t, m] = x[t + m*T] for t in [0, T), T = N/8.  Returns V[t, m] = X[t + m*T]."""
    T = v.shape[0]
    N = 8 * T
    v = v.astype(np.complex128).copy()
    t = np.arange(T)
    Ns = 1
    for R in RADICES[N]:
        nb = 8 // R  # butterflies per lane
        out = np.zeros(N, dtype=np.complex128)
        last = Ns * R == N
        for q in range(nb):
            j = t + q * T
            slots = [q + r * nb for r in range(R)]
            a = v[:, slots]  # [T, R]
            k = j % Ns
            ang = -2.0 * np.pi * k / (Ns * R)
            tw = np.exp(1j * ang[:, None] * np.arange(R)[None, :])
            a = a * tw
            u = np.fft.fft(a, axis=1)  # radix-R butterfly, natural order
            j0 = (j // Ns) * Ns * R + k
            if last:
                v[:, slots] = u
            else:
                for r in range(R):
                    out[j0 + r * Ns] = u[:, r]
        if not last:
            v = out.reshape(8, T).T.copy()  # v[t, m] = out[t + m*T]
        Ns *= R
    return v


def first_pass(img):
    """K_A: FFT along y (axis 0) of the real image, two columns per complex FFT.  Returns the
    intermediate H[ky in 0..N/2-1][x] complex with row 0 = F1[0,x] + i F1[N/2,x]."""
    N = img.shape[0]
    T = N // 8
    H = np.zeros((N // 2, N), dtype=np.complex128)
    t = np.arange(T)
    for xa in range(0, N, 2):
        z = img[:, xa] + 1j * img[:, xa + 1]
        v = z.reshape(8, T).T  # v[t, m] = z[t + m*T]
        V = fft_lanes(v)
        Z = np.zeros(N, dtype=np.complex128)
        for m in range(8):
            Z[t + m * T] = V[:, m]
        for m in range(4):
            k = t + m * T
            Zm = Z[(N - k) % N]
            A = 0.5 * (Z[k] + np.conj(Zm))
            B = -0.5j * (Z[k] - np.conj(Zm))
            H[k, xa] = A
            H[k, xa + 1] = B
        # lane 0: packed row
        H[0, xa] = Z[0].real + 1j * Z[N // 2].real
        H[0, xa + 1] = Z[0].imag + 1j * Z[N // 2].imag
    return H


def second_pass(H):
    """K_B: complex FFT along x of every intermediate row; un-packs row 0 into ky=0 and ky=N/2.
    Returns F[ky in 0..N/2][kx in 0..N-1]."""
    Nh, N = H.shape
    T = N // 8
    t = np.arange(T)
    F = np.zeros((Nh + 1, N), dtype=np.complex128)
    for ky in range(Nh):
        v = H[ky].reshape(8, T).T
        V = fft_lanes(v)
        C = np.zeros(N, dtype=np.complex128)
        for m in range(8):
            C[t + m * T] = V[:, m]
        if ky == 0:
            k = np.arange(N)
            Cm = C[(N - k) % N]
            F[0] = 0.5 * (C + np.conj(Cm))
            F[Nh] = -0.5j * (C - np.conj(Cm))
        else:
            F[ky] = C
    return F


def half_plane_weights(mask_shifted):
    """W[ky in 0..N/2][kx in 0..N-1] such that sum over the full fftshifted plane of mask*f equals
    sum over the half plane of W*f for any f with f(k) = f(-k)."""
    N = mask_shifted.shape[0]
    mu = np.fft.ifftshift(mask_shifted.astype(np.float64))
    ky = np.arange(N // 2 + 1)
    kx = np.arange(N)
    W = mu[ky[:, None], kx[None, :]].copy()
    mir = mu[(-ky[:, None]) % N, (-kx[None, :]) % N]
    inner = (ky > 0) & (ky < N // 2)
    W[inner] += mir[inner]
    return W


def pearson_from_moments(q, e, w):
    """Weighted Pearson over the half plane == masked Pearson over the full plane."""
    sw = w.sum()
    ebar = (w * e).sum() / sw
    wec = w * (e - ebar)
    s1 = (w * q).sum()
    s2 = (w * q * q).sum()
    s3 = (wec * q).sum()
    var_q = s2 - s1 * s1 / sw
    var_e = (w * (e - ebar) ** 2).sum()
    cov = s3 - (s1 / sw) * wec.sum()
    den = var_q * var_e
    if not den > 0:
        return 0.0
    return cov / np.sqrt(den)


def run_table(n, apix, twist, csym, rot, diameter, ball_radius, imax, tail_bits=24):
    """k_run_table: T[i + imax][ky] = sum over the csym copies s of G_(i,s)[ky], ky < N/2, with
    G_c[ky] = sum_y ey_c(y) exp(-2 pi i ky y / N) over the truncated rows of subunit c's footprint; slot
    ky = 0 packs (G[0], G[N/2]).  One subunit per asymmetric unit (radius diameter/2, azimuth 0, axial 0)."""
    sigma2 = ball_radius ** 2 / np.log(2.0)
    rpx = max(1, int(np.ceil(np.sqrt(sigma2 * tail_bits * np.log(2.0)) / apix)))
    y = np.arange(n)
    tab = np.zeros((2 * imax + 1, n // 2), dtype=np.complex128)
    for i in range(-imax, imax + 1):
        g_full = np.zeros(n, dtype=np.complex128)
        for s in range(csym):
            ang = np.deg2rad(rot + twist * i + 360.0 * s / csym)
            yc = diameter / 2.0 * np.sin(ang)  # row coordinate: does not depend on the rise
            cy = yc / apix + n // 2
            rows = y[np.abs(y - cy) <= rpx]
            ey = np.exp(-(((rows - n // 2) * apix - yc) ** 2) / sigma2)
            g_full += (ey[None, :] * np.exp(-2j * np.pi * np.outer(np.arange(n), rows) / n)).sum(axis=1)
        tab[i + imax] = g_full[: n // 2]
        tab[i + imax, 0] = g_full[0].real + 1j * g_full[n // 2].real
    return tab, rpx, sigma2


def first_pass_from_table(tab, rpx, sigma2, n, apix, rise, imax_t):
    """k_first_pass_table / the panel build of k_fused_pass: H[ky][x] = sum_i ex_i(x) T[i][ky] with the column
    factor ex_i(x) = exp(-dx^2 / sigma^2) inside the truncation window around the axial coordinate i * rise."""
    H = np.zeros((n // 2, n), dtype=np.complex128)
    imax = int(np.ceil(n * apix / rise))  # utils.py:153
    for i in range(-imax, imax + 1):
        xc = i * rise
        cx = xc / apix + n // 2
        cols = np.arange(n)[np.abs(np.arange(n) - cx) <= rpx]
        ex = np.exp(-(((cols - n // 2) * apix - xc) ** 2) / sigma2)
        H[:, cols] += tab[i + imax_t][:, None] * ex[None, :]
    return H
