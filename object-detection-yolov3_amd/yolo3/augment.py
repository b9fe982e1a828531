"""Host-side augmentation of (image, boxes) pairs, mirroring the reference's augment.py
(augment_image_box_pair :30-125, augment_boxes :128-189, apply_affine_transformation_boxes :192-272,
apply_affine_transformation :275-297, crop_to_size :20-27).

CPU pre-processing in the reader processes, as in the reference; it is stochastic (draws from the global np.random in
the reference's order) and outside the measured path.  scikit-image is not available in this image, so
the rescale of augment.py:277-280 is restated on scipy.ndimage.map_coordinates (rescale_bilinear).  Pinned by
tests/golden/augment.npz, produced by running the reference's augment.py under seeded np.random: boxes and crop
offsets identical, pixels to 1e-4 of the 8-bit range.  Boxes are [n,5] = x, y, w, h, class (top-left corner).
"""
import numpy as np
import scipy.ndimage

MIN_VISIBLE = 12      # boxes with less than 12 px inside the crop are dropped (augment.py:228-237)


def jitter_boxes(boxes, location_jitter, size_jitter, img_shape):
    """augment.py:128-189.  Returns None for an empty input (Q14)."""
    if boxes is None or boxes.shape[0] == 0:
        return None
    b = boxes.astype(np.int64).copy()
    x, y, w, h = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    for i in range(len(x)):
        x[i] += int(location_jitter * w[i] * np.random.randn())
        y[i] += int(location_jitter * h[i] * np.random.randn())
    for i in range(len(x)):
        d = int(size_jitter * w[i] * np.random.randn())
        x[i] -= int(d / 2)
        w[i] += d
        d = int(size_jitter * h[i] * np.random.randn())
        y[i] -= int(d / 2)
        h[i] += d
    x1 = np.minimum(x + w - 1, img_shape[1] - 1)
    y1 = np.minimum(y + h - 1, img_shape[0] - 1)
    x0, y0 = np.maximum(x, 0), np.maximum(y, 0)
    ww, hh = x1 - x0 + 1, y1 - y0 + 1
    assert np.all(hh > 0) and np.all(ww > 0), 'box with zero or negative size'      # augment.py:180, same failure mode
    return np.stack([x0, y0, ww, hh, b[:, 4]], 1).astype(np.int32)


def transform_boxes(boxes, crop_size, reflect_x, reflect_y, scale_x, scale_y, dx, dy):
    """augment.py:192-272: scale, shift by the crop origin, drop boxes that left the crop, clamp, reflect."""
    if boxes is None or boxes.shape[0] == 0:
        return None
    cls = boxes[:, 4]
    x0 = boxes[:, 0] * scale_x - dx
    x1 = (boxes[:, 0] + boxes[:, 2] - 1) * scale_x - dx
    y0 = boxes[:, 1] * scale_y - dy
    y1 = (boxes[:, 1] + boxes[:, 3] - 1) * scale_y - dy
    h, w = crop_size[0], crop_size[1]
    keep = ~((x0 >= w) | (y0 >= h) | (x1 < 0) | (y1 < 0))
    keep &= ~((x0 >= w - MIN_VISIBLE) | (y0 >= h - MIN_VISIBLE) | (x1 < MIN_VISIBLE) | (y1 < MIN_VISIBLE))
    if not keep.any():
        return None
    x0, y0, x1, y1, cls = x0[keep], y0[keep], x1[keep], y1[keep], cls[keep]
    x0, y0 = np.maximum(x0, 0), np.maximum(y0, 0)
    x1, y1 = np.minimum(x1, w - 1), np.minimum(y1, h - 1)
    if reflect_x:
        x0, x1 = w - x1, w - x0
    if reflect_y:
        y0, y1 = h - y1, h - y0
    return np.stack([x0, y0, x1 - x0 + 1, y1 - y0 + 1, cls], 1).astype(np.int32)


def rescale_bilinear(img, scale_y, scale_x):
    """skimage.transform.rescale(img, [scale_y, scale_x(, 1)], mode='reflect', preserve_range=True) as used at
    augment.py:277-280 (scikit-image is not in this image): output shape round(shape * scale); output pixel centres map to
    input coordinates (r + 0.5) * in_rows / out_rows - 0.5 (likewise for columns); bilinear; out-of-range samples reflect
    without repeating the edge (numpy.pad 'reflect' = scipy 'mirror').  Returns float64 like skimage does."""
    img = np.asarray(img, dtype=np.float64)
    rows = int(np.round(scale_y * img.shape[0]))
    cols = int(np.round(scale_x * img.shape[1]))
    r = (np.arange(rows) + 0.5) * (img.shape[0] / rows) - 0.5
    c = (np.arange(cols) + 0.5) * (img.shape[1] / cols) - 0.5
    grid = np.meshgrid(r, c, indexing='ij')
    if img.ndim == 2:
        return scipy.ndimage.map_coordinates(img, grid, order=1, mode='mirror')
    return np.stack([scipy.ndimage.map_coordinates(img[:, :, ch], grid, order=1, mode='mirror') for ch in range(img.shape[2])], axis=2)


def transform_image(img, reflect_x, reflect_y, scale_x, scale_y, crop_to):
    """augment.py:275-297: rescale, random crop to crop_to, flips.  Returns (img float64, dx, dy)."""
    if scale_x != 1 or scale_y != 1:
        img = rescale_bilinear(img, scale_y, scale_x)
    else:
        img = np.asarray(img, dtype=np.float64)      # skimage's identity rescale also returns float64 (values unchanged to 1e-12)
    dy = dx = 0
    if img.shape[0] - crop_to[0] > 0:
        dy = int(np.random.randint(0, img.shape[0] - crop_to[0]))
    if img.shape[1] - crop_to[1] > 0:
        dx = int(np.random.randint(0, img.shape[1] - crop_to[1]))
    img = img[dy:dy + crop_to[0], dx:dx + crop_to[1]]
    if reflect_x:
        img = np.fliplr(img)
    if reflect_y:
        img = np.flipud(img)
    return img, dx, dy


def crop_to_size(img, boxes, crop_to):
    """augment.py:20-27."""
    img, dx, dy = transform_image(img, False, False, 1.0, 1.0, crop_to)
    return img, transform_boxes(boxes, crop_to, False, False, 1.0, 1.0, dx, dy)


def augment_image_box_pair(img, boxes, rotation_flag=False, reflection_flag=False, crop_to=None, noise_augmentation_severity=0,
                           scale_augmentation_severity=0, blur_augmentation_max_sigma=0, box_size_augmentation_severity=0,
                           box_location_jitter_severity=0):
    """augment.py:30-125 (same parameter names and ranges)."""
    assert not rotation_flag, 'Rotation not implemented for image and boxes pair'
    img = np.asarray(img, dtype=np.float32)
    assert img.ndim in (2, 3)
    noise = noise_augmentation_severity or 0
    scale = scale_augmentation_severity or 0
    blur = blur_augmentation_max_sigma or 0
    assert 0 <= noise < 1 and 0 <= scale < 1 and 0 <= (box_size_augmentation_severity or 0) < 1 and 0 <= (box_location_jitter_severity or 0) < 1
    if crop_to is None:
        crop_to = img.shape[:2]
    reflect_x = reflect_y = False
    scale_x = scale_y = 1
    if reflection_flag:
        reflect_x = np.random.rand() > 0.5
        reflect_y = np.random.rand() > 0.5
    if scale > 0:
        hi = 1.0 + scale
        lo = max(max(crop_to[0] / img.shape[0], crop_to[1] / img.shape[1]), 1.0 - scale)
        scale_x = lo + (hi - lo) * np.random.rand()
        scale_y = lo + (hi - lo) * np.random.rand()
    boxes = jitter_boxes(boxes, box_location_jitter_severity or 0, box_size_augmentation_severity or 0, img.shape)
    img, dx, dy = transform_image(img, reflect_x, reflect_y, scale_x, scale_y, crop_to)
    boxes = transform_boxes(boxes, crop_to, reflect_x, reflect_y, scale_x, scale_y, dx, dy)
    if noise > 0:
        smax = noise * (np.max(img) - np.min(img))
        sigma = -smax + 2 * smax * np.random.rand()
        img = img + np.random.standard_normal(img.shape) * sigma
    if blur > 0:
        sigma = -blur + 2 * blur * np.random.rand()
        if sigma > 0:
            img = scipy.ndimage.gaussian_filter(img, sigma, mode='reflect')
    return np.asarray(img, dtype=np.float32), boxes
