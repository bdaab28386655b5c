"""CPU restatement of the reference's ZBP loader -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).

Follows beamformer_simple_parameters_from_zbp_file (tests/throughput.c:150-374) field by
field on top of the record layouts of external/zemp_bp.h:98-198, and writes files of those
layouts for the tests.  PARITY UNPINNED against the compiled reference: tests/throughput.c
needs <zstd.h>, which this image does not ship, so the harness cannot be built here; the
reference holds no .bp fixtures either.  The product loader (csrc/zbp.cpp) is checked against
this restatement on synthetic files of both header versions.
"""
import math
import struct

import numpy as np

MAGIC = 0x5042504D455AFECA                       # zemp_bp.h:23
V1 = np.dtype([                                   # zemp_bp.h:98-121
    ("magic", "<u8"), ("version", "<u4"), ("decode_mode", "<i2"), ("beamform_mode", "<i2"),
    ("raw_data_dimension", "<u4", 4), ("sample_count", "<u4"), ("channel_count", "<u4"),
    ("receive_event_count", "<u4"), ("frame_count", "<u4"), ("element_pitch", "<f4", 2),
    ("transform", "<f4", 16), ("channel_mapping", "<i2", 256), ("steering_angles", "<f4", 256),
    ("focal_depths", "<f4", 256), ("sparse_elements", "<i2", 256), ("hadamard_rows", "<i2", 256),
    ("speed_of_sound", "<f4"), ("demodulation_frequency", "<f4"), ("sampling_frequency", "<f4"),
    ("time_offset", "<f4"), ("transmit_mode", "<u4"), ("_pad", "<u4")])
V2 = np.dtype([                                   # zemp_bp.h:123-151
    ("magic", "<u8"), ("major", "<u4"), ("minor", "<u4"), ("raw_data_dimension", "<u4", 4),
    ("raw_data_kind", "<i4"), ("raw_data_offset", "<i4"), ("raw_data_compression_kind", "<i4"),
    ("decode_mode", "<i4"), ("sampling_mode", "<i4"), ("sampling_frequency", "<f4"),
    ("demodulation_frequency", "<f4"), ("speed_of_sound", "<f4"), ("channel_mapping_offset", "<i4"),
    ("sample_count", "<u4"), ("channel_count", "<u4"), ("receive_event_count", "<u4"),
    ("transform", "<f4", 16), ("element_pitch", "<f4", 2), ("time_offset", "<f4"),
    ("group_acquisition_time", "<f4"), ("ensemble_repetition_interval", "<f4"),
    ("acquisition_mode", "<i4"), ("acquisition_parameters_offset", "<i4"), ("contrast_mode", "<i4"),
    ("contrast_parameters_offset", "<i4"), ("emission_descriptors_offset", "<i4")])
assert V1.itemsize == 3728 and V2.itemsize == 184

KIND_BYTES = [2, 4, 4, 8, 2, 4]
FORCES, UFORCES, HERCULES, RCA_VLS, RCA_TPW, UHERCULES = 0, 1, 2, 3, 4, 5
SAMPLING_2X, SAMPLING_4X = 0, 1


def parameters(raw):
    """dict of the BeamformerSimpleParameters fields the reference's loader sets, plus
    'payload' = (kind, compression, offset, size); raises ValueError where it returns 0."""
    raw = bytes(raw)
    if len(raw) < 16 or struct.unpack_from("<Q", raw)[0] != MAGIC:
        raise ValueError("not a ZBP file")
    major = struct.unpack_from("<I", raw, 8)[0]
    f32 = np.float32
    bp = {}
    if major == 1:                                                  # :158-224
        h = np.frombuffer(raw, V1, 1)[0]
        C, A = int(h["channel_count"]), int(h["receive_event_count"])
        bp.update(sample_count=int(h["sample_count"]), channel_count=C, acquisition_count=A,
                  sampling_mode=SAMPLING_4X, acquisition_kind=int(h["beamform_mode"]),
                  decode_mode=int(h["decode_mode"]), sampling_frequency=f32(h["sampling_frequency"]),
                  demodulation_frequency=f32(h["sampling_frequency"]) / f32(4),
                  speed_of_sound=f32(h["speed_of_sound"]), time_offset=f32(h["time_offset"]),
                  channel_mapping=h["channel_mapping"][:C].copy(), xdc_transform=h["transform"].copy(),
                  xdc_element_pitch=h["element_pitch"].copy(),
                  raw_data_dimensions=h["raw_data_dimension"][:2].copy(), data_kind=0)
        mode = int(h["transmit_mode"])
        if mode >= 4:
            raise ValueError("transmit mode")
        bp["transmit_receive_orientation"] = [0x11, 0x12, 0x21, 0x22][mode]
        kind = bp["acquisition_kind"]
        if kind in (FORCES, HERCULES, UFORCES, UHERCULES):
            bp.update(single_focus=1, single_orientation=1,
                      focal_vector=np.array([h["steering_angles"][0], h["focal_depths"][0]], f32))
        if kind in (UFORCES, UHERCULES):
            bp["sparse_elements"] = h["sparse_elements"][:A].copy()
        if kind in (RCA_TPW, RCA_VLS):
            bp.update(focal_depths=h["focal_depths"][:A].copy(), steering_angles=h["steering_angles"][:A].copy(),
                      transmit_receive_orientations=np.full(A, bp["transmit_receive_orientation"], np.uint8))
        bp["emission"] = ("sine", f32(2), bp["demodulation_frequency"])
        bp["payload"] = (0, 1, 0, 0)
        return bp
    if major != 2:
        raise ValueError("version")
    h = np.frombuffer(raw, V2, 1)[0]                                # :226-366
    C, A = int(h["channel_count"]), int(h["receive_event_count"])
    bp.update(sample_count=int(h["sample_count"]), channel_count=C, acquisition_count=A,
              sampling_mode=[SAMPLING_4X, SAMPLING_2X][int(h["sampling_mode"])],
              acquisition_kind=int(h["acquisition_mode"]), decode_mode=int(h["decode_mode"]),
              sampling_frequency=f32(h["sampling_frequency"]), demodulation_frequency=f32(h["demodulation_frequency"]),
              speed_of_sound=f32(h["speed_of_sound"]), time_offset=f32(h["time_offset"]),
              contrast_mode=int(h["contrast_mode"]), xdc_transform=h["transform"].copy(),
              xdc_element_pitch=h["element_pitch"].copy(), raw_data_dimensions=h["raw_data_dimension"][:2].copy(),
              data_kind=int(h["raw_data_kind"]))
    off = int(h["channel_mapping_offset"])
    bp["channel_mapping"] = (np.frombuffer(raw, "<i2", C, off).copy() if off != -1 else np.arange(C, dtype=np.int16))
    payload = (int(h["raw_data_kind"]), int(h["raw_data_compression_kind"]), 0, 0)
    off = int(h["raw_data_offset"])
    if off != -1:
        if payload[1] == 1:
            size = len(raw) - off
        else:
            size = int(np.prod(h["raw_data_dimension"].astype(np.uint64))) * KIND_BYTES[payload[0]]
        payload = payload[:2] + (off, size)
    bp["payload"] = payload
    kind, poff = struct.unpack_from("<ii", raw, int(h["emission_descriptors_offset"]))
    if kind == 0:
        bp["emission"] = ("sine",) + tuple(np.frombuffer(raw, "<f4", 2, poff))
    elif kind == 1:
        bp["emission"] = ("chirp",) + tuple(np.frombuffer(raw, "<f4", 3, poff))
    else:
        raise ValueError("emission kind")
    ap = int(h["acquisition_parameters_offset"])
    mode = bp["acquisition_kind"]

    def focus():
        depth, angle, _origin, orientation = struct.unpack_from("<fffI", raw, ap)
        bp.update(transmit_receive_orientation=orientation, focal_vector=np.array([angle, depth], f32),
                  single_focus=1, single_orientation=1)

    def sparse():
        soff = struct.unpack_from("<i", raw, ap + 16)[0]
        bp["sparse_elements"] = np.frombuffer(raw, "<i2", A, soff).copy()

    if mode == FORCES:
        pass
    elif mode == HERCULES:
        focus()
    elif mode == UFORCES:
        sparse()
    elif mode == UHERCULES:
        focus(); sparse()
    elif mode == RCA_TPW:
        angles, orient = struct.unpack_from("<ii", raw, ap)
        bp.update(transmit_receive_orientations=np.frombuffer(raw, np.uint8, A, orient).copy(),
                  steering_angles=np.frombuffer(raw, "<f4", A, angles).copy(),
                  focal_depths=np.full(A, np.inf, f32))
    elif mode == RCA_VLS:                                           # :340-363
        d_off, o_off, orient = struct.unpack_from("<iii", raw, ap)
        depth = np.frombuffer(raw, "<f4", A, d_off)
        origin = np.frombuffer(raw, "<f4", A, o_off)
        sign = np.where(depth < 0, f32(-1), f32(1)).astype(f32)
        angle = np.array([f32(f32(f32(math.atan2(float(o), float(-d))) * f32(180.0)) / f32(3.14159265358979323846))
                          for o, d in zip(origin, depth)], f32)
        bp.update(transmit_receive_orientations=np.frombuffer(raw, np.uint8, A, orient).copy(),
                  steering_angles=angle,
                  focal_depths=(sign * np.sqrt(depth * depth + origin * origin, dtype=f32)).astype(f32))
    else:
        raise ValueError("acquisition mode")
    return bp


# ---------------------------------------------------------------- writers (tests only)

def write_v1(acquisition_kind, decode_mode, dims, samples, channels, events, pitch, transform, channel_mapping,
             steering_angles, focal_depths, sparse_elements, speed_of_sound, sampling_frequency, time_offset,
             transmit_mode):
    h = np.zeros(1, V1)[0]
    h["magic"], h["version"] = MAGIC, 1
    h["decode_mode"], h["beamform_mode"] = decode_mode, acquisition_kind
    h["raw_data_dimension"] = list(dims) + [1] * (4 - len(dims))
    h["sample_count"], h["channel_count"], h["receive_event_count"], h["frame_count"] = samples, channels, events, 1
    h["element_pitch"], h["transform"] = pitch, transform
    for name, values in (("channel_mapping", channel_mapping), ("steering_angles", steering_angles),
                         ("focal_depths", focal_depths), ("sparse_elements", sparse_elements)):
        h[name][: len(values)] = values
    h["speed_of_sound"], h["sampling_frequency"], h["time_offset"] = speed_of_sound, sampling_frequency, time_offset
    h["demodulation_frequency"] = sampling_frequency / 4
    h["transmit_mode"] = transmit_mode
    return h.tobytes()


def write_v2(acquisition_kind, data_kind, decode_mode, sampling_mode, dims, samples, channels, events, pitch, transform,
             speed_of_sound, sampling_frequency, demodulation_frequency, time_offset, emission, channel_mapping=None,
             focus=None, sparse_elements=None, tilting_angles=None, orientations=None, focal_depths=None,
             origin_offsets=None, data=None, compressed=False, contrast_mode=0):
    """Header followed by 4-byte-aligned records; offsets are patched as records are placed."""
    h = np.zeros(1, V2)[0]
    h["magic"], h["major"], h["minor"] = MAGIC, 2, 0
    h["raw_data_dimension"] = list(dims) + [1] * (4 - len(dims))
    h["raw_data_kind"], h["decode_mode"], h["sampling_mode"] = data_kind, decode_mode, sampling_mode
    h["sampling_frequency"], h["demodulation_frequency"], h["speed_of_sound"] = sampling_frequency, demodulation_frequency, speed_of_sound
    h["sample_count"], h["channel_count"], h["receive_event_count"] = samples, channels, events
    h["transform"], h["element_pitch"], h["time_offset"] = transform, pitch, time_offset
    h["acquisition_mode"], h["contrast_mode"] = acquisition_kind, contrast_mode
    h["contrast_parameters_offset"] = -1
    body = bytearray()

    def place(blob):
        while (V2.itemsize + len(body)) % 4:
            body.append(0)
        offset = V2.itemsize + len(body)
        body.extend(blob)
        return offset

    h["channel_mapping_offset"] = place(np.asarray(channel_mapping, "<i2").tobytes()) if channel_mapping is not None else -1
    if emission[0] == "sine":
        p = place(struct.pack("<ff", *emission[1:]))
        h["emission_descriptors_offset"] = place(struct.pack("<ii", 0, p))
    else:
        p = place(struct.pack("<fff", *emission[1:]))
        h["emission_descriptors_offset"] = place(struct.pack("<ii", 1, p))
    sparse_off = place(np.asarray(sparse_elements, "<i2").tobytes()) if sparse_elements is not None else 0
    if acquisition_kind in (HERCULES, UHERCULES):
        depth, angle, origin, orientation = focus
        rec = struct.pack("<fffI", depth, angle, origin, orientation)
        if acquisition_kind == UHERCULES:
            rec += struct.pack("<i", sparse_off)
        h["acquisition_parameters_offset"] = place(rec)
    elif acquisition_kind == UFORCES:
        h["acquisition_parameters_offset"] = place(struct.pack("<fffIi", 0, 0, 0, 0, sparse_off))
    elif acquisition_kind == RCA_TPW:
        a = place(np.asarray(tilting_angles, "<f4").tobytes())
        o = place(np.asarray(orientations, np.uint8).tobytes())
        h["acquisition_parameters_offset"] = place(struct.pack("<ii", a, o))
    elif acquisition_kind == RCA_VLS:
        d = place(np.asarray(focal_depths, "<f4").tobytes())
        g = place(np.asarray(origin_offsets, "<f4").tobytes())
        o = place(np.asarray(orientations, np.uint8).tobytes())
        h["acquisition_parameters_offset"] = place(struct.pack("<iii", d, g, o))
    else:
        h["acquisition_parameters_offset"] = -1
    h["raw_data_compression_kind"] = 1 if compressed else 0
    h["raw_data_offset"] = place(bytes(data)) if data is not None else -1      # payload last: zstd runs to EOF
    return h.tobytes() + bytes(body)


def zstd_compress(data):
    """The system libzstd (runtime only in this image), for building compressed test files."""
    import ctypes as C
    z = C.CDLL("libzstd.so.1")
    z.ZSTD_compressBound.restype = C.c_size_t
    z.ZSTD_compressBound.argtypes = [C.c_size_t]
    z.ZSTD_compress.restype = C.c_size_t
    z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
    data = bytes(data)
    out = C.create_string_buffer(z.ZSTD_compressBound(len(data)))
    n = z.ZSTD_compress(out, len(out), data, len(data), 3)
    return out.raw[:n]
