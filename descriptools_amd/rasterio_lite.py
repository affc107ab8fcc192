"""Headless GeoTIFF read / write for the rasters either side of the hot path.

The reference's example uses rasterio (`Example/example.py:33-39` to read, `:201-217` to write); it is not
in this image, PIL is.  `read` returns the first band as a numpy array plus the georeferencing tags;
`write` stores a 2-D array (uint8 / int16 / int32 / float32) carrying those tags over, so that an output
such as the classified flood map lines up with its inputs in a GIS.  No reprojection, no overviews.
"""
import numpy as np

# TIFF / GeoTIFF tags copied from a template raster
MODEL_PIXEL_SCALE, MODEL_TIEPOINT, GEO_KEY_DIRECTORY, GEO_DOUBLE_PARAMS, GEO_ASCII_PARAMS, GDAL_NODATA = (
    33550, 33922, 34735, 34736, 34737, 42113)
GEO_TAGS = (MODEL_PIXEL_SCALE, MODEL_TIEPOINT, GEO_KEY_DIRECTORY, GEO_DOUBLE_PARAMS, GEO_ASCII_PARAMS)
_TAG_TYPE = {MODEL_PIXEL_SCALE: 12, MODEL_TIEPOINT: 12, GEO_KEY_DIRECTORY: 3, GEO_DOUBLE_PARAMS: 12,
             GEO_ASCII_PARAMS: 2, GDAL_NODATA: 2}  # TIFF field types: 12 DOUBLE, 3 SHORT, 2 ASCII


def read(path):
    """-> (array[H, W], meta) with meta = {"tags": {tag: value}, "nodata": float | None, "pixel": (dx, dy)}."""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    with Image.open(path) as im:
        a = np.array(im)
        tags = {t: im.tag_v2[t] for t in GEO_TAGS if t in im.tag_v2}
        nod = im.tag_v2.get(GDAL_NODATA)
    nodata = None
    if nod is not None:
        try:
            nodata = float(str(nod).strip("\x00 "))
        except ValueError:
            nodata = None
    scale = tags.get(MODEL_PIXEL_SCALE)
    pixel = (float(scale[0]), float(scale[1])) if scale else None
    return a, {"tags": tags, "nodata": nodata, "pixel": pixel}


def read_masked(path, nodata_value=-100, dtype=None):
    """The reference's convention (`example.py:42-43`): cells equal to the raster's nodata (or, for float
    rasters, below -1e30) become `nodata_value`; optional cast."""
    a, meta = read(path)
    nod = meta["nodata"]
    mask = np.zeros(a.shape, bool)
    if a.dtype.kind == "f":
        mask |= a < -1e30
    if nod is not None and np.isfinite(nod) and abs(nod) < 1e30:
        mask |= a == nod
    a = np.where(mask, nodata_value, a)
    return (a.astype(dtype) if dtype is not None else a), meta


def write(path, array, like=None, nodata=None, compression="tiff_lzw"):
    """Write a 2-D array; `like` = meta from `read` (georeferencing tags are copied)."""
    from PIL import Image, TiffImagePlugin
    a = np.ascontiguousarray(array)
    if a.ndim != 2:
        raise ValueError("2-D array expected")
    mode = {"uint8": "L", "int16": "I;16S", "int32": "I", "float32": "F"}.get(a.dtype.name)
    if mode is None:
        raise ValueError("unsupported dtype %s (uint8, int16, int32, float32)" % a.dtype)
    if mode == "I;16S":  # PIL has no signed 16-bit writer: widen (exact)
        a, mode = a.astype(np.int32), "I"
    im = Image.fromarray(a, mode=mode)
    ifd = TiffImagePlugin.ImageFileDirectory_v2()
    for t, v in (like or {}).get("tags", {}).items():
        ifd.tagtype[t] = _TAG_TYPE[t]
        ifd[t] = v
    if nodata is not None:
        ifd.tagtype[GDAL_NODATA] = 2
        ifd[GDAL_NODATA] = repr(nodata) if isinstance(nodata, float) else str(nodata)
    im.save(path, format="TIFF", tiffinfo=ifd, compression=compression)
