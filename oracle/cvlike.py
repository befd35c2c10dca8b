"""TEST INFRASTRUCTURE ONLY -- CPU oracle, never imported by the product path.

numpy restatements of the third-party image operations the reference's hot
path calls but which are NOT under /root/reference and NOT installed in this
image (OpenCV-python 3.2.0, imutils, pyclipper 1.0.6; reference README.md:41-54):

    cv2.getPerspectiveTransform   reference common/generator.py:129
    cv2.warpPerspective(CUBIC)    reference common/generator.py:130-131
    imutils.rotate_bound          reference common/generator.py:163
    cv2.flip(., 0)                reference common/generator.py:165
    cv2.resize(INTER_AREA)        reference common/generator.py:169
    pyclipper + cv2.fillConvexPoly reference common/bad_weather.py:367-389

PARITY UNPINNED for everything in this file: the reference ships no tests or
golden vectors and the libraries cannot be imported here, so each function
restates the library's *published* algorithm from the OpenCV 3.x imgproc
sources as documented (1/32-pixel coordinate quantisation, float32 weight
tables, cubic A=-0.75, BORDER_CONSTANT 0, double accumulation for CV_64F).
Where a library's behaviour cannot be reproduced without its source
(Clipper's polygon output order, the SVD inside getPerspectiveTransform)
the rule used instead is stated in the docstring; those rules only influence
the per-drop colour constant (tolerance +-1 LSB on rainy_image), never the
alpha geometry.

All arithmetic is IEEE double (or float32 where OpenCV uses float tables) with
an explicit evaluation order, so that the HIP kernels can follow the same order
and be bit-exact against this file.
"""
import numpy as np

INTER_BITS = 5
INTER_TAB_SIZE = 1 << INTER_BITS          # 32
AB_BITS = 10
AB_SCALE = 1 << AB_BITS                   # 1024
INT_MIN = -2147483648
INT_MAX = 2147483647


def cv_round(v):
    """saturate_cast<int>(double): round-half-even then saturate (cvRound)."""
    r = np.rint(np.asarray(v, dtype=np.float64))
    r = np.where(r < INT_MIN, float(INT_MIN), r)
    r = np.where(r > INT_MAX, float(INT_MAX), r)
    r = np.where(np.isnan(r), float(INT_MIN), r)
    return r.astype(np.int64)


def sat_short(v):
    return np.clip(v, -32768, 32767)


# --------------------------------------------------------------------------
# weight tables (OpenCV imgwarp.cpp initInterTab1D / initInterTab2D, float)
# --------------------------------------------------------------------------
def cubic_tab_1d():
    """interpolateCubic(i/32) for i in 0..31, in float32 arithmetic, A=-0.75."""
    A = np.float32(-0.75)
    one = np.float32(1)
    tab = np.zeros((INTER_TAB_SIZE, 4), np.float32)
    scale = np.float32(1.0) / np.float32(INTER_TAB_SIZE)
    for i in range(INTER_TAB_SIZE):
        x = np.float32(i) * scale
        c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
        c1 = ((A + np.float32(2)) * x - (A + np.float32(3))) * x * x + one
        xm = one - x
        c2 = ((A + np.float32(2)) * xm - (A + np.float32(3))) * xm * xm + one
        c3 = one - c0 - c1 - c2
        tab[i] = (c0, c1, c2, c3)
    return tab


_CUBIC_1D = cubic_tab_1d()


def linear_tab_1d():
    tab = np.zeros((INTER_TAB_SIZE, 2), np.float32)
    scale = np.float32(1.0) / np.float32(INTER_TAB_SIZE)
    for i in range(INTER_TAB_SIZE):
        x = np.float32(i) * scale
        tab[i] = (np.float32(1) - x, x)
    return tab


_LINEAR_1D = linear_tab_1d()


# --------------------------------------------------------------------------
# getPerspectiveTransform
# --------------------------------------------------------------------------
def solve8(A, b):
    """Gaussian elimination, partial pivoting (first max |pivot| wins), explicit
    order.  OpenCV 3.2 uses an SVD solve here; the systems on this path are
    well-conditioned (eps=1e-3 in warping_points prevents singular quads), the two
    differ by O(1e-15) relative.  UNPINNED."""
    A = np.array(A, dtype=np.float64)
    b = np.array(b, dtype=np.float64)
    n = 8
    for col in range(n):
        piv = col
        best = abs(A[col, col])
        for r in range(col + 1, n):
            if abs(A[r, col]) > best:
                best = abs(A[r, col])
                piv = r
        if piv != col:
            A[[col, piv]] = A[[piv, col]]
            b[[col, piv]] = b[[piv, col]]
        with np.errstate(all='ignore'):
            for r in range(col + 1, n):
                f = A[r, col] / A[col, col]
                for c in range(col + 1, n):
                    A[r, c] = A[r, c] - f * A[col, c]
                b[r] = b[r] - f * b[col]
    x = np.zeros(n)
    with np.errstate(all='ignore'):
        for r in range(n - 1, -1, -1):
            s = b[r]
            for c in range(r + 1, n):
                s = s - A[r, c] * x[c]
            x[r] = s / A[r, r]
    return x


def get_perspective_transform(src, dst):
    """cv::getPerspectiveTransform(const Point2f src[4], const Point2f dst[4])."""
    src = np.asarray(src, np.float32).astype(np.float64)
    dst = np.asarray(dst, np.float32).astype(np.float64)
    A = np.zeros((8, 8))
    b = np.zeros(8)
    for i in range(4):
        A[i, 0] = A[i + 4, 3] = src[i, 0]
        A[i, 1] = A[i + 4, 4] = src[i, 1]
        A[i, 2] = A[i + 4, 5] = 1.0
        A[i, 6] = -src[i, 0] * dst[i, 0]
        A[i, 7] = -src[i, 1] * dst[i, 0]
        A[i + 4, 6] = -src[i, 0] * dst[i, 1]
        A[i + 4, 7] = -src[i, 1] * dst[i, 1]
        b[i] = dst[i, 0]
        b[i + 4] = dst[i, 1]
    x = solve8(A, b)
    return np.array([[x[0], x[1], x[2]], [x[3], x[4], x[5]], [x[6], x[7], 1.0]])


def invert3(M):
    """cv::invert for a 3x3 CV_64F matrix (closed form, DECOMP_LU branch)."""
    m = np.asarray(M, np.float64)
    d = (m[0, 0] * (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1])
         - m[0, 1] * (m[1, 0] * m[2, 2] - m[1, 2] * m[2, 0])
         + m[0, 2] * (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]))
    if d == 0.0 or not np.isfinite(d):
        return np.zeros((3, 3))
    d = 1.0 / d
    t = np.zeros(9)
    t[0] = (m[1, 1] * m[2, 2] - m[1, 2] * m[2, 1]) * d
    t[1] = (m[0, 2] * m[2, 1] - m[0, 1] * m[2, 2]) * d
    t[2] = (m[0, 1] * m[1, 2] - m[0, 2] * m[1, 1]) * d
    t[3] = (m[1, 2] * m[2, 0] - m[1, 0] * m[2, 2]) * d
    t[4] = (m[0, 0] * m[2, 2] - m[0, 2] * m[2, 0]) * d
    t[5] = (m[0, 2] * m[1, 0] - m[0, 0] * m[1, 2]) * d
    t[6] = (m[1, 0] * m[2, 1] - m[1, 1] * m[2, 0]) * d
    t[7] = (m[0, 1] * m[2, 0] - m[0, 0] * m[2, 1]) * d
    t[8] = (m[0, 0] * m[1, 1] - m[0, 1] * m[1, 0]) * d
    return t.reshape(3, 3)


# --------------------------------------------------------------------------
# warpPerspective, INTER_CUBIC, BORDER_CONSTANT(0), CV_64F, one channel
# --------------------------------------------------------------------------
def warp_perspective_cubic(src, M, dw, dh):
    """cv::warpPerspective(src, M, (dw,dh), INTER_CUBIC) on a 1-channel f64 image.

    WarpPerspectiveInvoker (32x32-ish blocks, coordinates * INTER_TAB_SIZE rounded
    with cvRound) followed by remapBicubic<Cast<double,double>, float, 1>."""
    src = np.asarray(src, np.float64)
    sh, sw = src.shape
    Mi = invert3(M).reshape(9)
    BLOCK_SZ = 32
    bh0 = min(BLOCK_SZ // 2, dh)
    bw0 = min(BLOCK_SZ * BLOCK_SZ // bh0, dw)
    bh0 = min(BLOCK_SZ * BLOCK_SZ // bw0, dh)
    xs = np.arange(dw)
    bx = (xs // bw0) * bw0
    x1 = (xs - bx).astype(np.float64)
    bxf = bx.astype(np.float64)[None, :]
    ys = np.arange(dh).astype(np.float64)[:, None]
    with np.errstate(all='ignore'):
        X0 = Mi[0] * bxf + Mi[1] * ys + Mi[2]
        Y0 = Mi[3] * bxf + Mi[4] * ys + Mi[5]
        W0 = Mi[6] * bxf + Mi[7] * ys + Mi[8]
        W = W0 + Mi[6] * x1[None, :]
        W = np.where(W != 0, float(INTER_TAB_SIZE) / np.where(W != 0, W, 1.0), 0.0)
        fX = np.maximum(float(INT_MIN), np.minimum(float(INT_MAX), (X0 + Mi[0] * x1[None, :]) * W))
        fY = np.maximum(float(INT_MIN), np.minimum(float(INT_MAX), (Y0 + Mi[3] * x1[None, :]) * W))
    X = cv_round(fX)
    Y = cv_round(fY)
    sx = sat_short(X >> INTER_BITS) - 1
    sy = sat_short(Y >> INTER_BITS) - 1
    fx = X & (INTER_TAB_SIZE - 1)
    fy = Y & (INTER_TAB_SIZE - 1)
    cx = _CUBIC_1D[fx]                      # (dh,dw,4) float32
    cy = _CUBIC_1D[fy]
    # 2-D table entries are float products vy*vx rounded to float32
    w2 = (cy[..., :, None] * cx[..., None, :]).astype(np.float32).astype(np.float64)  # (dh,dw,4,4)

    width1 = max(sw - 3, 0)
    height1 = max(sh - 3, 0)
    interior = (sx >= 0) & (sx < width1) & (sy >= 0) & (sy < height1)
    # gather taps with zero substitution outside the image
    taps = np.zeros((dh, dw, 4, 4))
    for i in range(4):
        yy = sy + i
        vy = (yy >= 0) & (yy < sh)
        yyc = np.clip(yy, 0, sh - 1)
        for j in range(4):
            xx = sx + j
            vx = (xx >= 0) & (xx < sw)
            xxc = np.clip(xx, 0, sw - 1)
            taps[:, :, i, j] = np.where(vy & vx, src[yyc, xxc], 0.0)
    p = taps * w2
    # interior path: per-row groups, then added row by row
    rows = [((p[..., i, 0] + p[..., i, 1]) + p[..., i, 2]) + p[..., i, 3] for i in range(4)]
    s_int = rows[0]
    for i in range(1, 4):
        s_int = s_int + rows[i]
    # border path: sum starts at cval*1 (=0) and adds in-range taps one by one
    s_bor = np.zeros((dh, dw))
    for i in range(4):
        for j in range(4):
            s_bor = s_bor + p[..., i, j]
    return np.where(interior, s_int, s_bor)


# --------------------------------------------------------------------------
# imutils.rotate_bound  (getRotationMatrix2D + warpAffine INTER_LINEAR)
# --------------------------------------------------------------------------
def rotation_matrix_2d(cx, cy, alpha, beta):
    """cv::getRotationMatrix2D(center, angle, 1) given alpha=cos(angle), beta=sin(angle);
    center is a Point2f (float32)."""
    cx = float(np.float32(cx))
    cy = float(np.float32(cy))
    return np.array([[alpha, beta, (1 - alpha) * cx - beta * cy],
                     [-beta, alpha, beta * cx + (1 - alpha) * cy]])


def rotate_bound_geometry(h, w, alpha, beta):
    """imutils.rotate_bound canvas and matrix: cos/sin of (-angle).

    Centre convention: `(cX, cY) = (w / 2, h / 2)` with TRUE division, as in the released imutils (>= 0.4,
    Python 3); very old copies used `w // 2, h // 2`, which differs by half a pixel for odd sizes and is not
    implemented (tests/test_known_answers.py pins the choice on a 33 x 229 canvas)."""
    cX, cY = w / 2, h / 2
    M = rotation_matrix_2d(cX, cY, alpha, beta)
    cos = abs(M[0, 0])
    sin = abs(M[0, 1])
    nW = int((h * sin) + (w * cos))
    nH = int((h * cos) + (w * sin))
    M[0, 2] += (nW / 2) - cX
    M[1, 2] += (nH / 2) - cY
    return M, nW, nH


def invert_affine(M):
    """The in-place inversion at the top of cv::warpAffine (no WARP_INVERSE_MAP)."""
    m = np.array(M, np.float64).reshape(6)
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11 = m[4] * D
    A22 = m[0] * D
    m[0] = A11
    m[1] = m[1] * (-D)
    m[3] = m[3] * (-D)
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2] = b1
    m[5] = b2
    return m


def warp_affine_linear(src, M, dw, dh):
    """cv::warpAffine(src, M, (dw,dh)) INTER_LINEAR, BORDER_CONSTANT 0, 1-channel f64:
    WarpAffineInvoker fixed-point coordinates (AB_BITS=10, round_delta=16) +
    remapBilinear<Cast<double,double>, RemapNoVec, float>."""
    src = np.asarray(src, np.float64)
    sh, sw = src.shape
    if dw <= 0 or dh <= 0:
        return np.zeros((max(dh, 0), max(dw, 0)))
    m = invert_affine(M)
    xs = np.arange(dw).astype(np.float64)
    ys = np.arange(dh).astype(np.float64)
    adelta = cv_round(m[0] * xs * AB_SCALE)
    bdelta = cv_round(m[3] * xs * AB_SCALE)
    round_delta = AB_SCALE // INTER_TAB_SIZE // 2
    X0 = cv_round((m[1] * ys + m[2]) * AB_SCALE) + round_delta
    Y0 = cv_round((m[4] * ys + m[5]) * AB_SCALE) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = sat_short(X >> INTER_BITS)
    sy = sat_short(Y >> INTER_BITS)
    fx = X & (INTER_TAB_SIZE - 1)
    fy = Y & (INTER_TAB_SIZE - 1)
    lx = _LINEAR_1D[fx]
    ly = _LINEAR_1D[fy]
    w = (ly[..., :, None] * lx[..., None, :]).astype(np.float32).astype(np.float64)   # (dh,dw,2,2)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < sh) & (xx >= 0) & (xx < sw)
        return np.where(ok, src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)], 0.0)

    v00 = tap(sy, sx)
    v01 = tap(sy, sx + 1)
    v10 = tap(sy + 1, sx)
    v11 = tap(sy + 1, sx + 1)
    return ((v00 * w[..., 0, 0] + v01 * w[..., 0, 1]) + v10 * w[..., 1, 0]) + v11 * w[..., 1, 1]


def rotate_bound(src, alpha, beta):
    """imutils.rotate_bound(src, angle) with alpha=cos(-angle*pi/180), beta=sin(-angle*pi/180)."""
    h, w = src.shape
    M, nW, nH = rotate_bound_geometry(h, w, alpha, beta)
    return warp_affine_linear(src, M, nW, nH)


def flip0(src):
    """cv2.flip(src, 0): vertical flip."""
    return src[::-1].copy()


# --------------------------------------------------------------------------
# cv2.resize(..., interpolation=INTER_AREA), CV_64F, one channel
# --------------------------------------------------------------------------
def _area_tab(ssize, dsize, scale):
    """computeResizeAreaTab: list of (si, di, alpha float32)."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1 = int(np.ceil(fsx1))
        sx2 = int(np.floor(fsx2))
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((sx1 - 1, dx, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((sx, dx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((sx2, dx, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _linear_area_coords(ssize, dsize, scale, inv_scale, is_x):
    """The area_mode branch of the INTER_LINEAR coefficient set-up in cv::resize."""
    ofs = np.zeros(dsize, np.int64)
    coef = np.zeros((dsize, 2), np.float32)
    xmax = dsize
    for d in range(dsize):
        s = int(np.floor(d * scale))
        f = np.float32((d + 1) - (s + 1) * inv_scale)
        f = np.float32(0) if f <= 0 else np.float32(f - np.float32(np.floor(f)))
        if is_x:
            if s < 0:
                f = np.float32(0)
                s = 0
            if s + 1 >= ssize:
                xmax = min(xmax, d)
                if s >= ssize - 1:
                    f = np.float32(0)
                    s = ssize - 1
        ofs[d] = s
        coef[d] = (np.float32(1) - f, f)
    return ofs, coef, xmax


def resize_area(src, dw, dh):
    """cv::resize(src, (dw,dh), interpolation=INTER_AREA) for 1-channel CV_64F."""
    src = np.asarray(src, np.float64)
    sh, sw = src.shape
    inv_sx = float(dw) / sw
    inv_sy = float(dh) / sh
    scale_x = 1.0 / inv_sx
    scale_y = 1.0 / inv_sy
    iscale_x = int(cv_round(scale_x))
    iscale_y = int(cv_round(scale_y))
    eps = np.finfo(np.float64).eps
    is_area_fast = abs(scale_x - iscale_x) < eps and abs(scale_y - iscale_y) < eps
    if scale_x >= 1 and scale_y >= 1:
        if is_area_fast:
            area = iscale_x * iscale_y
            scale = np.float32(1.0) / np.float32(area)          # float scale = 1.f/(area)
            dst = np.zeros((dh, dw))
            taps = [(ky, kx) for ky in range(iscale_y) for kx in range(iscale_x)]
            ys = np.arange(dh) * iscale_y
            xs = np.arange(dw) * iscale_x
            s = np.zeros((dh, dw))
            k = 0
            while k <= area - 4:
                q = [src[np.ix_(ys + taps[k + t][0], xs + taps[k + t][1])] for t in range(4)]
                s = s + (((q[0] + q[1]) + q[2]) + q[3])
                k += 4
            while k < area:
                s = s + src[np.ix_(ys + taps[k][0], xs + taps[k][1])]
                k += 1
            dst[:] = s * float(scale)
            return dst
        xtab = _area_tab(sw, dw, scale_x)
        ytab = _area_tab(sh, dh, scale_y)
        dst = np.zeros((dh, dw))
        xsi = np.array([t[0] for t in xtab])
        xdi = np.array([t[1] for t in xtab])
        xal = np.array([float(t[2]) for t in xtab])
        # horizontal pass for every source row: buf[sy][dx] += S[sy][si]*alpha, in tab order
        buf = np.zeros((sh, dw))
        for k in range(len(xtab)):
            buf[:, xdi[k]] = buf[:, xdi[k]] + src[:, xsi[k]] * xal[k]
        # vertical pass in tab order
        prev_dy = -1
        acc = None
        for (sy, dy, beta) in ytab:
            b = float(beta)
            if dy != prev_dy:
                if prev_dy >= 0:
                    dst[prev_dy] = acc
                acc = b * buf[sy]
                prev_dy = dy
            else:
                acc = acc + b * buf[sy]
        if prev_dy >= 0:
            dst[prev_dy] = acc
        return dst
    # dst larger than src along some axis: bilinear with "area" coordinates
    xofs, alpha, xmax = _linear_area_coords(sw, dw, scale_x, inv_sx, True)
    yofs, beta, _ = _linear_area_coords(sh, dh, scale_y, inv_sy, False)
    a0 = alpha[:, 0].astype(np.float64)[None, :]
    a1 = alpha[:, 1].astype(np.float64)[None, :]
    x0 = xofs
    x1 = np.minimum(xofs + 1, sw - 1)
    hres = src[:, x0] * a0 + src[:, x1] * a1
    if xmax < dw:
        hres[:, xmax:] = src[:, x0[xmax:]] * 1.0
    r0 = np.clip(yofs, 0, sh - 1)
    r1 = np.clip(yofs + 1, 0, sh - 1)
    b0 = beta[:, 0].astype(np.float64)[:, None]
    b1 = beta[:, 1].astype(np.float64)[:, None]
    return hres[r0] * b0 + hres[r1] * b1


# --------------------------------------------------------------------------
# FOV polygon rasterisation (replaces pyclipper intersection + fillConvexPoly)
# --------------------------------------------------------------------------
def polygon_to_int(pts):
    """pyclipper casts path coordinates to its integer type: truncation toward zero."""
    return np.trunc(np.asarray(pts, np.float64)).astype(np.int64)


def _div_round_half_up(num, den):
    """round(num/den) to nearest, ties toward +inf, exact integer arithmetic, den>0."""
    return (2 * num + den) // (2 * den)


def fov_rowspans(poly_int, rows, cols):
    """Row spans of the FOV polygon inside the rows x cols map.

    Rule (ours; the reference goes pyclipper-intersection -> cv2.fillConvexPoly,
    neither importable here -- UNPINNED): for every pixel row y that the polygon's
    y-extent touches, the filled span is [min, max] over all polygon edges that reach
    row y of the edge's x at that row (nearest integer, ties up; horizontal edges
    contribute both end points), clamped to [0, cols-1].  This is what
    fillConvexPoly's left/right edge walk produces for the x-monotone polygons
    compute_fov_plane_points emits (20-gon, or 24-gon hanging from the top/bottom
    border), up to boundary-pixel conventions.

    Returns (y0, xl[], xr[]) with xl>xr marking an empty row; y0 is the first row.
    """
    P = np.asarray(poly_int, np.int64)
    n = len(P)
    ymin = int(P[:, 1].min())
    ymax = int(P[:, 1].max())
    ya = max(ymin, 0)
    yb = min(ymax, rows - 1)
    if n < 3 or ya > yb:
        return 0, np.zeros(0, np.int64), np.zeros(0, np.int64)
    ys = np.arange(ya, yb + 1)
    big = np.int64(1) << 40
    xl = np.full(len(ys), big)
    xr = np.full(len(ys), -big)
    for i in range(n):
        x0, y0 = P[i]
        x1, y1 = P[(i + 1) % n]
        lo, hi = (y0, y1) if y0 <= y1 else (y1, y0)
        sel = (ys >= lo) & (ys <= hi)
        if not sel.any():
            continue
        if y0 == y1:
            cmin, cmax = min(x0, x1), max(x0, x1)
            xl[sel] = np.minimum(xl[sel], cmin)
            xr[sel] = np.maximum(xr[sel], cmax)
        else:
            if y1 < y0:                       # orient so den > 0
                xa, yA, xb, yB = x1, y1, x0, y0
            else:
                xa, yA, xb, yB = x0, y0, x1, y1
            den = yB - yA
            num = (xb - xa) * (ys[sel] - yA)
            xv = xa + _div_round_half_up(num, den)
            xl[sel] = np.minimum(xl[sel], xv)
            xr[sel] = np.maximum(xr[sel], xv)
    xl = np.maximum(xl, 0)
    xr = np.minimum(xr, cols - 1)
    return ya, xl, xr


def fill_fov_mask(mask, poly_int):
    """mask[...] = 1 inside the FOV row spans (the oracle's stand-in for
    cv2.fillConvexPoly(mask, s, 1) at reference common/bad_weather.py:388)."""
    rows, cols = mask.shape
    y0, xl, xr = fov_rowspans(poly_int, rows, cols)
    for k in range(len(xl)):
        if xl[k] <= xr[k]:
            mask[y0 + k, xl[k]:xr[k] + 1] = 1
    return mask
