"""Minimal reader and writer for Erdas Imagine HFA (.img) single-band rasters -- enough for HiPIMS model directories.

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


# ------------------------------------------------------------------------------------------------ writer
# The reference writes its outputs through GDAL with the driver the XML names -- format="HFA" in its example model
# (src/Datasets/CRasterDataset.cpp:101-183: one band, GDT_Float64, geotransform = top-left corner, rows top to bottom).
# The writer below produces the same thing without GDAL: an Ehfa tree with one Eimg_Layer of f64 pixels in uncompressed
# 64 x 64 blocks (GDAL's HFA block size), its RasterDMS block table, the Ehfa_Layer record naming the block type, and an
# Eprj_MapInfo with the pixel-centre coordinates.  The MIF dictionary lists exactly the types used, in the published
# syntax of the format.  Not written: statistics, pyramids, projection and the no-data node (the value stays -9999 in
# the data, as in the reference's rasters).
_DICTIONARY = (
    "{1:lversion,1:LfreeList,1:LrootEntryPtr,1:sentryHeaderLength,1:LdictionaryPtr,}Ehfa_File,"
    "{1:Lnext,1:Lprev,1:Lparent,1:Lchild,1:Ldata,1:ldataSize,64:cname,32:ctype,1:tmodTime,}Ehfa_Entry,"
    "{16:clabel,1:LheaderPtr,}Ehfa_HeaderTag,"
    "{1:LfreeList,1:lfreeSize,}Ehfa_FreeListNode,"
    "{1:lsize,1:Lptr,}Ehfa_Data,"
    "{1:lwidth,1:lheight,1:e3:thematic,athematic,fft of real-valued data,layerType,"
    "1:e13:u1,u2,u4,u8,s8,u16,s16,u32,s32,f32,f64,c64,c128,pixelType,1:lblockWidth,1:lblockHeight,}Eimg_Layer,"
    "{1:e2:raster,vector,type,1:LdictionaryPtr,}Ehfa_Layer,"
    "{1:sfileCode,1:Loffset,1:lsize,1:e2:false,true,logvalid,"
    "1:e2:no compression,ESRI GRID compression,compressionType,}Edms_VirtualBlockInfo,"
    "{1:lmin,1:lmax,}Edms_FreeIDList,"
    "{1:lnumvirtualblocks,1:lnumobjectsperblock,1:lnextobjectnum,"
    "1:e2:no compression,RLC compression,compressionType,"
    "0:poEdms_VirtualBlockInfo,blockinfo,0:poEdms_FreeIDList,freelist,1:tmodTime,}Edms_State,"
    "{0:pcstring,}Emif_String,"
    "{1:dx,1:dy,}Eprj_Coordinate,"
    "{1:dwidth,1:dheight,}Eprj_Size,"
    "{0:pcproName,1:*oEprj_Coordinate,upperLeftCenter,1:*oEprj_Coordinate,lowerRightCenter,"
    "1:*oEprj_Size,pixelSize,0:pcunits,}Eprj_MapInfo,"
    "."
)


def write_hfa(path, north_up, upper_left_corner=(0.0, 0.0), pixel_size=1.0, units="meters", block=64):
    """Write `north_up` (array[rows, cols], row 0 = north) as a one-band f64 HFA raster; `upper_left_corner` is the outer
    corner of the top-left pixel (a GDAL geotransform's origin), as the reference sets it."""
    a = np.ascontiguousarray(north_up, dtype=np.float64)
    rows, cols = a.shape
    bw = bh = int(block)
    bx, by = (cols + bw - 1) // bw, (rows + bh - 1) // bh
    nblocks, block_bytes = bx * by, bw * bh * 8
    padded = np.full((by * bh, bx * bw), -9999.0)
    padded[:rows, :cols] = a

    buf = bytearray(b"EHFA_HEADER_TAG\0" + struct.pack("<I", 20))
    buf += b"\0" * 18                                        # Ehfa_File, filled in at the end

    def reserve(n):
        off = len(buf)
        buf.extend(b"\0" * n)
        return off

    def entry(name, typ):
        return dict(off=reserve(128), name=name, type=typ, data=0, size=0, next=0, prev=0, parent=0, child=0)

    root, layer = entry("root", "root"), entry("Layer_1", "Eimg_Layer")
    dms, elayer, mapinfo = entry("RasterDMS", "Edms_State"), entry("Ehfa_Layer", "Ehfa_Layer"), entry("Map_Info", "Eprj_MapInfo")
    root["child"] = layer["off"]
    layer["parent"] = root["off"]; layer["child"] = dms["off"]
    chain = [dms, elayer, mapinfo]
    for i, e in enumerate(chain):
        e["parent"] = layer["off"]
        e["prev"] = chain[i - 1]["off"] if i else 0
        e["next"] = chain[i + 1]["off"] if i + 1 < len(chain) else 0

    # the pixel blocks, then the node data that points at them
    blocks_off = reserve(nblocks * block_bytes)
    for i in range(nblocks):
        r, c = divmod(i, bx)
        buf[blocks_off + i * block_bytes: blocks_off + (i + 1) * block_bytes] = \
            np.ascontiguousarray(padded[r * bh:(r + 1) * bh, c * bw:(c + 1) * bw]).tobytes()

    layer["data"] = reserve(20); layer["size"] = 20
    struct.pack_into("<iihhii", buf, layer["data"], cols, rows, 1, 10, bw, bh)       # athematic, f64

    dms["size"] = 14 + 8 + 14 * nblocks + 8 + 4
    dms["data"] = reserve(dms["size"])
    struct.pack_into("<iiih", buf, dms["data"], nblocks, bw * bh, nblocks * bw * bh, 0)
    struct.pack_into("<II", buf, dms["data"] + 14, nblocks, dms["data"] + 22)
    for i in range(nblocks):
        struct.pack_into("<hIihh", buf, dms["data"] + 22 + 14 * i, 0, blocks_off + i * block_bytes, block_bytes, 1, 0)
    struct.pack_into("<III", buf, dms["data"] + 22 + 14 * nblocks, 0, 0, 0)            # empty free list, modTime

    layer_dict = ("{%d:ddata,}RasterDMS,." % (bw * bh)).encode() + b"\0"            # the type of one block: bw*bh doubles
    layer_dict_off = reserve(len(layer_dict))
    buf[layer_dict_off:layer_dict_off + len(layer_dict)] = layer_dict
    elayer["data"] = reserve(6); elayer["size"] = 6
    struct.pack_into("<hI", buf, elayer["data"], 0, layer_dict_off)

    pro, un = b"Unknown\0", units.encode() + b"\0"
    mapinfo["size"] = 8 + len(pro) + 3 * 24 + 8 + len(un)
    d = mapinfo["data"] = reserve(mapinfo["size"])
    struct.pack_into("<II", buf, d, len(pro), d + 8)
    buf[d + 8:d + 8 + len(pro)] = pro
    d2 = d + 8 + len(pro)
    ulx, uly = upper_left_corner[0] + 0.5 * pixel_size, upper_left_corner[1] - 0.5 * pixel_size
    struct.pack_into("<IIdd", buf, d2, 1, d2 + 8, ulx, uly)
    struct.pack_into("<IIdd", buf, d2 + 24, 1, d2 + 32, ulx + (cols - 1) * pixel_size, uly - (rows - 1) * pixel_size)
    struct.pack_into("<IIdd", buf, d2 + 48, 1, d2 + 56, pixel_size, pixel_size)
    struct.pack_into("<II", buf, d2 + 72, len(un), d2 + 80)
    buf[d2 + 80:d2 + 80 + len(un)] = un

    dict_off = reserve(len(_DICTIONARY) + 1)
    buf[dict_off:dict_off + len(_DICTIONARY)] = _DICTIONARY.encode()
    for e in (root, layer, dms, elayer, mapinfo):
        struct.pack_into("<IIIIIi", buf, e["off"], e["next"], e["prev"], e["parent"], e["child"], e["data"], e["size"])
        buf[e["off"] + 24:e["off"] + 24 + len(e["name"])] = e["name"].encode()
        buf[e["off"] + 88:e["off"] + 88 + len(e["type"])] = e["type"].encode()
    struct.pack_into("<iIIhI", buf, 20, 1, 0, root["off"], 128, dict_off)
    with open(path, "wb") as f:
        f.write(bytes(buf))


def write_raster(path, south_up, resolution, origin=(0.0, 0.0)):
    """The reference's convention (CRasterDataset.cpp:166-172): `origin` is the bottom-left corner of the domain."""
    a = np.asarray(south_up, dtype=np.float64)
    write_hfa(path, a[::-1], (origin[0], origin[1] + resolution * a.shape[0]), resolution)
