"""ctypes binding of the C++ host mirror (point-cloud-donkey_amd/libism3d_amd.so, capi_host.cpp) for the tests."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "point-cloud-donkey_amd", "libism3d_amd.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        import torch  # noqa: F401  (one HIP runtime per process, see capi.lib)
        C.CDLL(os.path.join(ROOT, "point-cloud-donkey_amd", "libismhip.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(LIB)
        L.ism3d_new.restype = C.c_void_p
        L.ism3d_last_error.restype = C.c_char_p
        L.ism3d_delete.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class HostError(RuntimeError):
    pass


def _f(a):
    return np.ascontiguousarray(np.asarray(a, np.float32))


def _p(a):
    return C.c_void_p(0) if a is None else C.c_void_p(a.ctypes.data)


class Model:
    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.ism3d_new())
        self.L.ism3d_set_logging(self.h, 0)

    def _ck(self, rc, what):
        if rc != 0:
            raise HostError(f"{what}: {self.L.ism3d_last_error().decode()} (rc {rc})")

    def read(self, path, training=False):
        self._ck(self.L.ism3d_read(self.h, path.encode(), int(training)), "readObject")

    def write(self, path):
        self._ck(self.L.ism3d_write(self.h, path.encode()), "writeObject")

    def config_from_json(self, text):
        self._ck(self.L.ism3d_config_from_json(self.h, text.encode()), "configFromJson")

    def config_to_json(self):
        buf = C.create_string_buffer(1 << 16)
        self._ck(self.L.ism3d_config_to_json(self.h, buf, len(buf)), "configToJson")
        return buf.value.decode()

    def add_training(self, xyz, normals, class_id, instance_id, rgba=None):
        x, y, z = (_f(xyz[:, i]) for i in range(3)); nx, ny, nz = (_f(normals[:, i]) for i in range(3))
        c = None if rgba is None else np.ascontiguousarray(rgba, np.uint32)
        self._ck(self.L.ism3d_add_training(self.h, len(x), _p(x), _p(y), _p(z), _p(nx), _p(ny), _p(nz), _p(c), class_id, instance_id), "addTrainingModel")

    def add_training_file(self, path, class_id, instance_id):
        self._ck(self.L.ism3d_add_training_file(self.h, path.encode(), class_id, instance_id), "addTrainingModel(file)")

    def train(self):
        self._ck(self.L.ism3d_train(self.h), "train")

    def codebook_size(self):
        return self.L.ism3d_codebook_size(self.h)

    def codebook(self, dim, n_classes):
        n = self.codebook_size()
        words = np.empty((n, dim), np.float32); vx = np.empty((n * 4, 3), np.float32); vc = np.empty(n * 4, np.uint32); sg = np.empty(n_classes, np.float32)
        nv = self.L.ism3d_codebook_get(self.h, _p(words), _p(vx), _p(vc), _p(sg))
        return words, vx[:nv], vc[:nv], sg

    def set_codebook(self, cb, n_classes):
        """cb: dict with words [n,dim], vote_offsets, vote_xyz, vote_class, vote_instance, class_sigma and optional word_id, word_class,
        word_weight, word_keypoint, vote_weight, vote_class_weight, vote_bbox_quat, vote_bbox_size"""
        u = lambda k: None if cb.get(k) is None else np.ascontiguousarray(cb[k], np.uint32)
        f = lambda k: None if cb.get(k) is None else _f(cb[k])
        words = _f(cb["words"])
        wid = None if cb.get("word_id") is None else np.ascontiguousarray(cb["word_id"], np.int32)
        self._ck(self.L.ism3d_codebook_set(self.h, words.shape[0], words.shape[1], _p(words), _p(wid), _p(u("word_class")), _p(f("word_weight")),
                                           _p(f("word_keypoint")), _p(u("vote_offsets")), _p(f("vote_xyz")), _p(f("vote_weight")), _p(f("vote_class_weight")),
                                           _p(u("vote_class")), _p(u("vote_instance")), _p(f("vote_bbox_quat")), _p(f("vote_bbox_size")), n_classes,
                                           _p(f("class_sigma"))), "setCodebookData")

    def codebook_all(self):
        nw, dim, nc = C.c_int(), C.c_int(), C.c_int()
        nv = self.L.ism3d_codebook_get_all(self.h, C.byref(nw), C.byref(dim), C.byref(nc), *([None] * 14))
        n, d, c = nw.value, dim.value, nc.value
        out = dict(words=np.zeros((n, d), np.float32), word_id=np.zeros(n, np.int32), word_class=np.zeros(n, np.uint32), word_weight=np.zeros(n, np.float32),
                   word_keypoint=np.zeros((n, 3), np.float32), vote_offsets=np.zeros(n + 1, np.uint32), vote_xyz=np.zeros((nv, 3), np.float32),
                   vote_weight=np.zeros(nv, np.float32), vote_class_weight=np.zeros(nv, np.float32), vote_class=np.zeros(nv, np.uint32),
                   vote_instance=np.zeros(nv, np.uint32), vote_bbox_quat=np.zeros((nv, 4), np.float32), vote_bbox_size=np.zeros((nv, 3), np.float32),
                   class_sigma=np.zeros(c, np.float32))
        keys = ["words", "word_id", "word_class", "word_weight", "word_keypoint", "vote_offsets", "vote_xyz", "vote_weight", "vote_class_weight", "vote_class",
                "vote_instance", "vote_bbox_quat", "vote_bbox_size", "class_sigma"]
        self.L.ism3d_codebook_get_all(self.h, None, None, None, *[_p(out[k]) for k in keys])
        return out

    def set_labels(self, class_labels, instance_labels, instance_to_class):
        ca = (C.c_char_p * len(class_labels))(*[s.encode() for s in class_labels])
        ia = (C.c_char_p * len(instance_labels))(*[s.encode() for s in instance_labels])
        m = np.ascontiguousarray(instance_to_class, np.uint32)
        self._ck(self.L.ism3d_set_labels(self.h, len(class_labels), ca, len(instance_labels), ia, _p(m)), "setLabels")

    def label(self, which, idx):
        buf = C.create_string_buffer(4096)
        n = self.L.ism3d_get_label(self.h, which, idx, buf, len(buf))
        return buf.value.decode() if n >= 0 else None

    def dimensions(self, class_id):
        out = np.zeros(4, np.float32)
        return out if self.L.ism3d_dimensions(self.h, class_id, _p(out)) == 0 else None

    def set_dimensions(self, class_id, radius, box, radius_var=0.0, box_var=0.0):
        self._ck(self.L.ism3d_set_dimensions(self.h, class_id, _p(np.asarray([radius, box, radius_var, box_var], np.float32))), "setDimensions")

    def detect_batch(self, pt_off, xyz, normals, max_maxima=8, rgba=None):
        po = np.ascontiguousarray(pt_off, np.uint32)
        n_obj = len(po) - 1
        x, y, z = (_f(xyz[:, i]) for i in range(3)); nx, ny, nz = (_f(normals[:, i]) for i in range(3))
        c = None if rgba is None else np.ascontiguousarray(rgba, np.uint32)
        n = np.empty(n_obj, np.int32); pos = np.empty((n_obj, max_maxima, 3), np.float32); w = np.empty((n_obj, max_maxima), np.float32)
        cls = np.empty((n_obj, max_maxima), np.int32); inst = np.empty((n_obj, max_maxima), np.int32); nv = np.empty((n_obj, max_maxima), np.int32)
        quat = np.empty((n_obj, max_maxima, 4), np.float32); n_total = np.empty(n_obj, np.int32)
        self._ck(self.L.ism3d_detect_batch(self.h, n_obj, _p(po), _p(x), _p(y), _p(z), _p(nx), _p(ny), _p(nz), _p(c), max_maxima, _p(n), _p(pos), _p(w),
                                           _p(cls), _p(inst), _p(nv), _p(quat), _p(n_total)), "detectBatch")
        return dict(n=n, pos=pos, weight=w, cls=cls, inst=inst, n_votes=nv, bbox_quat=quat, n_total=n_total)

    def detect_file(self, path):
        cls, w = C.c_int32(), C.c_float()
        self._ck(self.L.ism3d_detect_file(self.h, path.encode(), C.byref(cls), C.byref(w)), "detect(file)")
        return cls.value, w.value

    def close(self):
        if self.h:
            self.L.ism3d_delete(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def load_cloud(path, cap=1 << 20):
    L = lib()
    x = np.empty(cap, np.float32); y = np.empty(cap, np.float32); z = np.empty(cap, np.float32)
    nx = np.empty(cap, np.float32); ny = np.empty(cap, np.float32); nz = np.empty(cap, np.float32); c = np.empty(cap, np.uint32)
    n = L.ism3d_load_cloud(path.encode(), cap, _p(x), _p(y), _p(z), _p(nx), _p(ny), _p(nz), _p(c))
    if n < 0:
        raise HostError(f"loadPointCloud: {L.ism3d_last_error().decode()} (rc {n})")
    return np.stack([x[:n], y[:n], z[:n]], 1), np.stack([nx[:n], ny[:n], nz[:n]], 1), c[:n]


def lzf_compress(data: bytes) -> bytes:
    """small greedy LZF encoder (test helper): literal runs and back references as liblzf / pcl::lzfCompress emit them"""
    out = bytearray(); lit = bytearray(); i = 0; n = len(data); table = {}

    def flush():
        nonlocal lit
        for s in range(0, len(lit), 32):
            chunk = lit[s:s + 32]
            out.append(len(chunk) - 1); out.extend(chunk)
        lit = bytearray()
    while i < n:
        key = data[i:i + 3]
        j = table.get(key, -1) if len(key) == 3 else -1
        if len(key) == 3:
            table[key] = i
        if j >= 0 and 0 < i - j <= 8192:
            ln = 3
            while i + ln < n and ln < 264 and data[j + ln] == data[i + ln]:
                ln += 1
            flush()
            dist = i - j - 1
            l2 = ln - 2
            if l2 < 7:
                out.append((l2 << 5) | (dist >> 8))
            else:
                out.append((7 << 5) | (dist >> 8)); out.append(l2 - 7)
            out.append(dist & 0xff)
            i += ln
        else:
            lit.append(data[i]); i += 1
    flush()
    return bytes(out)


def write_pcd_compressed(path, xyz, normals, rgba=None):
    """PCD DATA binary_compressed as pcl::PCDWriter::writeBinaryCompressed lays it out (field-major plain stream, LZF)"""
    import struct
    n = len(xyz)
    fields = ["x", "y", "z"] + (["rgb"] if rgba is not None else []) + ["normal_x", "normal_y", "normal_z", "curvature"]
    types = ["F", "F", "F"] + (["U"] if rgba is not None else []) + ["F", "F", "F", "F"]
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT 1\n"
           "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary_compressed\n") % (" ".join(fields), " ".join(["4"] * len(fields)), " ".join(types),
                                                                            " ".join(["1"] * len(fields)), n, n)
    cols = [xyz[:, 0], xyz[:, 1], xyz[:, 2]]
    if rgba is not None:
        cols.append(rgba.astype(np.uint32).view(np.float32))
    cols += [normals[:, 0], normals[:, 1], normals[:, 2], np.zeros(n, np.float32)]
    plain = b"".join(np.ascontiguousarray(c, np.float32).tobytes() for c in cols)
    comp = lzf_compress(plain)
    with open(path, "wb") as f:
        f.write(hdr.encode()); f.write(struct.pack("<II", len(comp), len(plain))); f.write(comp)


def write_pcd(path, xyz, normals, rgba=None, binary=False, height=1):
    n = len(xyz)
    assert n % height == 0
    fields = ["x", "y", "z"] + (["rgb"] if rgba is not None else []) + ["normal_x", "normal_y", "normal_z", "curvature"]
    types = ["F", "F", "F"] + (["U"] if rgba is not None else []) + ["F", "F", "F", "F"]
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS %s\nSIZE %s\nTYPE %s\nCOUNT %s\nWIDTH %d\nHEIGHT %d\n"
           "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n") % (" ".join(fields), " ".join(["4"] * len(fields)), " ".join(types),
                                                              " ".join(["1"] * len(fields)), n // height, height, n, "binary" if binary else "ascii")
    with open(path, "wb") as f:
        f.write(hdr.encode())
        if binary:
            cols = [xyz.astype(np.float32)]
            if rgba is not None:
                cols.append(rgba.astype(np.uint32).view(np.float32).reshape(-1, 1))
            cols += [normals.astype(np.float32), np.zeros((n, 1), np.float32)]
            f.write(np.ascontiguousarray(np.concatenate(cols, axis=1)).tobytes())
        else:
            for i in range(n):
                row = ["%.9g" % v for v in xyz[i]]
                if rgba is not None:
                    row.append(str(int(rgba[i])))
                row += ["%.9g" % v for v in normals[i]] + ["0"]
                f.write((" ".join(row) + "\n").encode())
