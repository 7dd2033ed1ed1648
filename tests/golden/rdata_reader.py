"""Minimal reader for R's XDR serialization (.RData, format RDX2/RDX3).

Dev-container-only helper used by the make_*_golden.py scripts in this
directory to turn the reference's own data fixtures
(/root/reference/workflow/data-objects/*.RData) into small committed golden
vectors.  It only understands the SEXP types those files contain.  It is not
part of the product and nothing under varscot_amd/ imports it.
"""
import gzip
import struct


class _Reader:
    def __init__(self, buf):
        self.b = buf
        self.p = 0
        self.refs = []

    def i32(self):
        v = struct.unpack_from(">i", self.b, self.p)[0]
        self.p += 4
        return v

    def f64s(self, n):
        v = struct.unpack_from(">%dd" % n, self.b, self.p)
        self.p += 8 * n
        return list(v)

    def i32s(self, n):
        v = struct.unpack_from(">%di" % n, self.b, self.p)
        self.p += 4 * n
        return list(v)

    def raw(self, n):
        v = self.b[self.p:self.p + n]
        self.p += n
        return v

    def length(self):
        n = self.i32()
        if n == -1:
            hi, lo = self.i32(), self.i32()
            n = (hi << 32) | (lo & 0xFFFFFFFF)
        return n

    def item(self):
        flags = self.i32()
        t = flags & 0xFF
        has_attr = bool(flags & (1 << 9))
        has_tag = bool(flags & (1 << 10))
        if t == 254:      # NILVALUE
            return None
        if t == 253:      # R_EmptyEnv
            return {"_env": "empty"}
        if t == 242:      # R_GlobalEnv
            return {"_env": "global"}
        if t == 255:      # REFSXP
            return self.refs[(flags >> 8) - 1]
        if t == 1:        # SYMSXP
            name = self.item()
            sym = ("sym", name)
            self.refs.append(sym)
            return sym
        if t in (2, 6):   # LISTSXP / LANGSXP: pairlist
            out = []
            while True:
                attr = self.item() if has_attr else None
                tag = self.item() if has_tag else None
                car = self.item()
                out.append((tag[1] if tag else None, car))
                nflags = struct.unpack_from(">i", self.b, self.p)[0]
                nt = nflags & 0xFF
                if nt == 254:
                    self.p += 4
                    break
                if nt != t:
                    # dotted tail
                    out.append((None, self.item()))
                    break
                flags = self.i32()
                has_attr = bool(flags & (1 << 9))
                has_tag = bool(flags & (1 << 10))
            return ("pairlist", out)
        if t == 9:        # CHARSXP
            n = self.i32()
            if n == -1:
                return None
            return self.raw(n).decode("latin-1")
        if t in (10, 13):  # LGLSXP / INTSXP
            n = self.length()
            val = self.i32s(n)
        elif t == 14:     # REALSXP
            n = self.length()
            val = self.f64s(n)
        elif t == 16:     # STRSXP
            n = self.length()
            val = [self.item() for _ in range(n)]
        elif t in (19, 20):  # VECSXP / EXPRSXP
            n = self.length()
            val = [self.item() for _ in range(n)]
        elif t == 24:     # RAWSXP
            n = self.length()
            val = self.raw(n)
        elif t == 3:      # CLOSXP
            attr = self.item() if has_attr else None
            env, formals, body = self.item(), self.item(), self.item()
            return ("closure", formals, body)
        elif t == 4:      # ENVSXP
            locked = self.i32()
            env = {"_env": "env"}
            self.refs.append(env)
            enclos, frame, hashtab, attr = self.item(), self.item(), self.item(), self.item()
            return env
        elif t in (7, 8):  # SPECIALSXP / BUILTINSXP
            n = self.i32()
            return ("builtin", self.raw(n).decode())
        else:
            raise ValueError("unsupported SEXP type %d at %d" % (t, self.p))
        attrs = {}
        if has_attr:
            a = self.item()
            if a:
                attrs = {k: v for k, v in a[1]}
        return {"type": t, "val": val, "attr": attrs}


def load_rdata(path):
    """Return {object name: parsed object} for an .RData file."""
    with gzip.open(path, "rb") as f:
        buf = f.read()
    assert buf[:5] in (b"RDX2\n", b"RDX3\n"), buf[:5]
    assert buf[5:7] == b"X\n"
    r = _Reader(buf)
    r.p = 7
    version = r.i32()
    r.i32()
    r.i32()
    if version == 3:
        n = r.i32()
        r.raw(n)
    top = r.item()
    return {k: v for k, v in top[1]}


def data_frame(obj):
    """VECSXP with names -> {column: python list}; factors become level strings."""
    names = obj["attr"]["names"]["val"]
    out = {}
    for name, col in zip(names, obj["val"]):
        vals = col["val"]
        lv = col["attr"].get("levels")
        if lv is not None:
            vals = [lv["val"][v - 1] if v is not None and v > 0 else None for v in vals]
        out[name] = vals
    return out
