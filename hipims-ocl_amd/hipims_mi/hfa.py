"""Minimal reader for Erdas Imagine HFA (.img) single-band rasters -- enough for HiPIMS model directories.

The reference reads rasters through GDAL (src/Datasets/CRasterDataset.cpp:101-183); GDAL is not available here, and
the only raster format the reference's example ships is an RLE-compressed float32 HFA file.  This reader walks the
Ehfa_Entry tree, finds the first Eimg_Layer and its RasterDMS (Edms_State) block table and decodes each block
(uncompressed, or "ESRI GRID compression": min value + bit-packed values, optionally run-length counted).
Rows are returned north-up exactly as stored; `read_raster` flips them south-up like the reference does on load
(CRasterDataset.cpp:411).
"""
from __future__ import annotations

import struct

import numpy as np

_PIXEL = {  # Eimg_Layer pixelType enum -> (numpy dtype, bits)
    0: (None, 1), 1: (None, 2), 2: (None, 4), 3: (np.uint8, 8), 4: (np.int8, 8), 5: (np.uint16, 16),
    6: (np.int16, 16), 7: (np.uint32, 32), 8: (np.int32, 32), 9: (np.float32, 32), 10: (np.float64, 64),
}


class HFAError(ValueError):
    pass


def _entry(data, off):
    nxt, prev, parent, child, dptr, dsize = struct.unpack_from("<IIIIIi", data, off)
    name = data[off + 24:off + 88].split(b"\0")[0].decode("latin1")
    typ = data[off + 88:off + 120].split(b"\0")[0].decode("latin1")
    return dict(next=nxt, child=child, data=dptr, size=dsize, name=name, type=typ)


def _walk(data, off):
    while off:
        e = _entry(data, off)
        yield e
        if e["child"]:
            yield from _walk_children(data, e)
        off = e["next"]


def _walk_children(data, parent):
    off = parent["child"]
    while off:
        e = _entry(data, off)
        e["parent"] = parent
        yield e
        if e["child"]:
            yield from _walk_children(data, e)
        off = e["next"]


def _read_bits(buf, start_byte, index, nbits):
    """Value `index` of a big-endian bit-packed array of `nbits`-wide unsigned integers."""
    if nbits == 0:
        return 0
    if nbits == 8:
        return buf[start_byte + index]
    if nbits == 16:
        return (buf[start_byte + 2 * index] << 8) | buf[start_byte + 2 * index + 1]
    if nbits == 32:
        b = start_byte + 4 * index
        return (buf[b] << 24) | (buf[b + 1] << 16) | (buf[b + 2] << 8) | buf[b + 3]
    bit = index * nbits                       # 1, 2, 4 bits: packed low bits first within a byte
    byte = buf[start_byte + (bit >> 3)]
    return (byte >> (bit & 7)) & ((1 << nbits) - 1)


def _decode_block(raw, npix, dtype, bits):
    """One 'ESRI GRID compression' block -> npix values of dtype (integers offset by the block minimum; floats are
    carried as their 32-bit patterns)."""
    data_min, num_runs, data_offset = struct.unpack_from("<IiI", raw, 0)
    nbits = raw[12]
    out = np.empty(npix, np.uint32 if bits <= 32 else np.uint64)
    if num_runs == -1:                        # bit-packed values, no run lengths
        for i in range(npix):
            out[i] = (_read_bits(raw, 13, i, nbits) + data_min) & 0xFFFFFFFF
    else:
        pos, produced = 13, 0
        for run in range(num_runs):
            b0 = raw[pos]
            count = b0 & 0x3F
            extra = b0 >> 6
            for k in range(extra):
                count = (count << 8) | raw[pos + 1 + k]
            pos += 1 + extra
            value = (_read_bits(raw, data_offset, run, nbits) + data_min) & 0xFFFFFFFF
            end = min(npix, produced + count)
            out[produced:end] = value
            produced = end
        if produced != npix:
            raise HFAError(f"RLE block decoded {produced} of {npix} pixels")
    if dtype in (np.float32,):
        return out.astype(np.uint32).view(np.float32)
    return out.astype(dtype)


def read_hfa(path):
    """Return (array[rows, cols] north-up, info dict) of the first raster layer in an HFA file."""
    data = open(path, "rb").read()
    if data[:15] != b"EHFA_HEADER_TAG":
        raise HFAError("not an HFA file")
    hdr = struct.unpack_from("<I", data, 16)[0]
    _version, _free, root, _ehl, _dict = struct.unpack_from("<iIIhI", data, hdr)
    layer = dms = mapinfo = None
    for e in _walk(data, root):
        if e["type"] == "Eimg_Layer" and layer is None:
            layer = e
        elif e["type"] == "Edms_State" and dms is None and layer is not None and e.get("parent") is layer:
            dms = e
        elif e["type"] == "Eprj_MapInfo" and mapinfo is None:
            mapinfo = e
    if layer is None or dms is None:
        raise HFAError("no raster layer found")
    width, height, _ltype, ptype, bw, bh = struct.unpack_from("<iihhii", data, layer["data"])
    dtype, bits = _PIXEL.get(ptype, (None, 0))
    if dtype is None:
        raise HFAError(f"unsupported pixel type {ptype}")
    # Edms_State: numvirtualblocks, numobjectsperblock, nextobjectnum, compressionType(e2 -> int16), blockinfo(count, ptr)
    nblocks, _nobj, _nextobj, _ctype = struct.unpack_from("<iiih", data, dms["data"])
    count, ptr = struct.unpack_from("<II", data, dms["data"] + 14)
    if count != nblocks:
        raise HFAError("block table size mismatch")
    bx, by = (width + bw - 1) // bw, (height + bh - 1) // bh
    out = np.zeros((by * bh, bx * bw), dtype)
    for i in range(nblocks):
        _fc, off, size, valid, comp = struct.unpack_from("<hIihh", data, ptr + 14 * i)
        r, c = divmod(i, bx)
        if not valid:
            continue
        if comp == 0:
            blk = np.frombuffer(data, dtype, bw * bh, off)
        else:
            blk = _decode_block(data[off:off + size], bw * bh, dtype, bits)
        out[r * bh:(r + 1) * bh, c * bw:(c + 1) * bw] = blk.reshape(bh, bw)
    info = dict(cols=width, rows=height, block=(bw, bh), pixel_type=ptype, blocks=nblocks)
    if mapinfo is not None:
        # Eprj_MapInfo: proName (Emif string ptr), upperLeftCenter (x,y), lowerRightCenter (x,y), pixelSize (w,h), units
        d = mapinfo["data"]
        n, p = struct.unpack_from("<II", data, d)
        d2 = d + 8 + n
        try:
            (n1, _p1) = struct.unpack_from("<II", data, d2)
            ulx, uly = struct.unpack_from("<dd", data, d2 + 8)
            (n2, _p2) = struct.unpack_from("<II", data, d2 + 24)
            lrx, lry = struct.unpack_from("<dd", data, d2 + 32)
            (n3, _p3) = struct.unpack_from("<II", data, d2 + 48)
            pw, ph = struct.unpack_from("<dd", data, d2 + 56)
            info.update(upper_left_center=(ulx, uly), lower_right_center=(lrx, lry), pixel_size=(pw, ph))
        except struct.error:
            pass
    return out[:height, :width], info


def read_raster(path):
    """Raster as the reference holds it: row 0 = south (CRasterDataset.cpp:411), float64."""
    arr, info = read_hfa(path)
    return np.ascontiguousarray(arr[::-1]).astype(np.float64), info
