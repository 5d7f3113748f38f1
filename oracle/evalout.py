"""CPU oracle of the eval output stage (TEST INFRASTRUCTURE ONLY -- tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this; the product path never does).

numpy/scipy restatement of the reference's host-side frame post-processing:
    frame_u8                (rgb_map.clamp(0,1).numpy()*255).astype('uint8')      TriPlane/main.py:98,117
    visualize_depth_numpy   TriPlane/utils.py:32-47  (the colour table is passed in: cv2 is absent; jet_lut restates OpenCV's
                            published Jet table, pinned to its printed entries in tests/test_evalout_oracle.py, not to a cv2 run)
    mse / psnr              TriPlane/main.py:105-106
    rgb_ssim                TriPlane/utils.py:109-155
Pinned by tests/golden/evalout.npz (outputs of the reference's own rgb_ssim / visualize_depth_numpy arithmetic run in
the build container, tests/golden/make_golden.py) and tests/test_evalout_oracle.py.
"""
import numpy as np
import scipy.signal


def frame_u8(rgb):
    x = np.clip(np.asarray(rgb, np.float32), np.float32(0), np.float32(1))
    return (x * 255).astype('uint8')


def depth_index(depth, minmax=None):
    """utils.py:37-45 up to (but not including) applyColorMap: the uint8 index image and [mi, ma]."""
    x = np.nan_to_num(np.asarray(depth, np.float32))
    if minmax is None:
        mi = np.min(x[x > 0])
        ma = np.max(x)
    else:
        mi, ma = minmax
    x = (x - mi) / (ma - mi + 1e-8)
    with np.errstate(invalid="ignore"):
        x = (255 * x).astype(np.uint8)
    return x, [mi, ma]


def visualize_depth_numpy(depth, minmax=None, lut=None):
    idx, rng = depth_index(depth, minmax)
    return lut[idx], rng


def mse(a, b):
    d = np.asarray(a, np.float32) - np.asarray(b, np.float32)
    return float(np.mean((d * d).astype(np.float64)))


def psnr(a, b):
    return -10.0 * np.log(mse(a, b)) / np.log(10.0)


def rgb_ssim(img0, img1, max_val, filter_size=11, filter_sigma=1.5, k1=0.01, k2=0.03, return_map=False):
    img0 = np.asarray(img0, np.float32)
    img1 = np.asarray(img1, np.float32)
    hw = filter_size // 2
    shift = (2 * hw - filter_size + 1) / 2
    f_i = ((np.arange(filter_size) - hw + shift) / filter_sigma) ** 2
    filt = np.exp(-0.5 * f_i)
    filt /= np.sum(filt)

    def blur(z):
        cols = []
        for i in range(z.shape[-1]):
            v = scipy.signal.convolve2d(z[..., i], filt[:, None], mode='valid')
            cols.append(scipy.signal.convolve2d(v, filt[None, :], mode='valid'))
        return np.stack(cols, -1)

    mu0, mu1 = blur(img0), blur(img1)
    mu00, mu11, mu01 = mu0 * mu0, mu1 * mu1, mu0 * mu1
    sigma00 = blur(img0 * img0) - mu00          # float32 products, as the reference squares torch float32 tensors
    sigma11 = blur(img1 * img1) - mu11
    sigma01 = blur(img0 * img1) - mu01
    sigma00 = np.maximum(0., sigma00)
    sigma11 = np.maximum(0., sigma11)
    sigma01 = np.sign(sigma01) * np.minimum(np.sqrt(sigma00 * sigma11), np.abs(sigma01))
    c1 = (k1 * max_val) ** 2
    c2 = (k2 * max_val) ** 2
    numer = (2 * mu01 + c1) * (2 * sigma01 + c2)
    denom = (mu00 + mu11 + c1) * (sigma00 + sigma11 + c2)
    ssim_map = numer / denom
    return ssim_map if return_map else float(np.mean(ssim_map))
