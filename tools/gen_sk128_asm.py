"""Generates e2e_multi_view_matching_amd/csrc/sinkhorn128_rows.h: the row pass of sinkhorn_resident128 / sinkhorn_resident2k as asm
statements on registers the compiler does not allocate (a K row in vector registers: v[B + 4 k + e], in accumulation registers:
a[B + 4 k + e]; k = 256-column chunk of the half, e = element of the lane's 4 adjacent columns; temporaries v56 - v63).

Every block exists in two forms: plain, and `_l` = the same block with the four ds_read_b128 of ONE LDS row (half) issued in front
of it and waited for behind it - the LDS latency of the rows that live in LDS disappears behind the block's multiply-adds.
Dependent packed operations are never adjacent."""
import os


def lit(lines):
    return "\n".join('        "%s\\n\\t"' % ln for ln in lines[:-1]) + '\n        "%s"' % lines[-1]


class Block:
    """Operand numbering: the block's own outputs, then (in the _l form) four 128-bit LDS destinations, then the inputs, then (in the
    _l form) the LDS address and the byte offset."""

    def __init__(self, n_out, n_in, with_lds):
        self.n_out, self.n_in, self.l = n_out, n_in, with_lds

    def o(self, i):  # logical operand index -> position
        return "%%%d" % (i if i < self.n_out else i + (4 if self.l else 0))

    def V(self, b, off):  # a pair of hidden vector registers
        return "v[%s+%d:%s+%d+1]" % (self.o(b), off, self.o(b), off)

    def wrap(self, lines):
        if not self.l:
            return lit(lines)
        t = ["%%%d" % (self.n_out + i) for i in range(4)]
        addr, off = "%%%d" % (self.n_out + 4 + self.n_in), "%%%d" % (self.n_out + 4 + self.n_in + 1)
        pre = ["ds_read_b128 %s, %s offset:%s+%d" % (t[i], addr, off, 1024 * i) for i in range(4)]
        return lit(pre + lines + ["s_waitcnt lgkmcnt(0)"])


def rs4v(l):  # acc 0-3 | blo 4-7, bhi 8-11, bases 12-15
    B = Block(4, 12, l)
    L = []
    for k in range(4):
        for q in range(4):
            if k == 0:
                L.append("v_pk_mul_f32 %s, %s, %s" % (B.o(q), B.V(12 + q, 0), B.o(4)))
            else:
                L.append("v_pk_fma_f32 %s, %s, %s, %s" % (B.o(q), B.V(12 + q, 4 * k), B.o(4 + k), B.o(q)))
        for q in range(4):
            L.append("v_pk_fma_f32 %s, %s, %s, %s" % (B.o(q), B.V(12 + q, 4 * k + 2), B.o(8 + k), B.o(q)))
    return B.wrap(L)


def rs2a(l):  # acc 0, 1 | blo 2-5, bhi 6-9, bases 10, 11
    B = Block(2, 10, l)
    L = []
    for k in range(4):
        for e in range(4):
            L.append("v_accvgpr_read_b32 v%d, a[%s+%d]" % (56 + e, B.o(10), 4 * k + e))
        for e in range(4):
            L.append("v_accvgpr_read_b32 v%d, a[%s+%d]" % (60 + e, B.o(11), 4 * k + e))
        if k == 0:
            L.append("v_pk_mul_f32 %s, v[56:57], %s" % (B.o(0), B.o(2 + k)))
            L.append("v_pk_mul_f32 %s, v[60:61], %s" % (B.o(1), B.o(2 + k)))
        else:
            L.append("v_pk_fma_f32 %s, v[56:57], %s, %s" % (B.o(0), B.o(2 + k), B.o(0)))
            L.append("v_pk_fma_f32 %s, v[60:61], %s, %s" % (B.o(1), B.o(2 + k), B.o(1)))
        L.append("v_pk_fma_f32 %s, v[58:59], %s, %s" % (B.o(0), B.o(6 + k), B.o(0)))
        L.append("v_pk_fma_f32 %s, v[62:63], %s, %s" % (B.o(1), B.o(6 + k), B.o(1)))
    return B.wrap(L)


def rc4(l, acc_rows):  # cl 0-3, ch 4-7 (in/out) | a2 8-11, bases 12-15   (not generated with LDS reads: 30 operands are the limit)
    B = Block(8, 8, l)
    L = []
    for q in range(4):
        for k in range(4):
            if acc_rows:
                t = 56 + 4 * (k & 1)
                for e in range(4):
                    L.append("v_accvgpr_read_b32 v%d, a[%s+%d]" % (t + e, B.o(12 + q), 4 * k + e))
                L.append("v_pk_fma_f32 %s, v[%d:%d], %s, %s" % (B.o(k), t, t + 1, B.o(8 + q), B.o(k)))
                L.append("v_pk_fma_f32 %s, v[%d:%d], %s, %s" % (B.o(4 + k), t + 2, t + 3, B.o(8 + q), B.o(4 + k)))
            else:
                L.append("v_pk_fma_f32 %s, %s, %s, %s" % (B.o(k), B.V(12 + q, 4 * k), B.o(8 + q), B.o(k)))
                L.append("v_pk_fma_f32 %s, %s, %s, %s" % (B.o(4 + k), B.V(12 + q, 4 * k + 2), B.o(8 + q), B.o(4 + k)))
    return B.wrap(L)


def rs2v(l):  # acc lo0 0, lo1 1, hi0 2, hi1 3 (four chains) | blo 4-7, bhi 8-11, bases 12, 13
    B = Block(4, 10, l)
    L = []
    for k in range(4):
        for q in range(2):
            if k == 0:
                L.append("v_pk_mul_f32 %s, %s, %s" % (B.o(q), B.V(12 + q, 0), B.o(4)))
            else:
                L.append("v_pk_fma_f32 %s, %s, %s, %s" % (B.o(q), B.V(12 + q, 4 * k), B.o(4 + k), B.o(q)))
        for q in range(2):
            if k == 0:
                L.append("v_pk_mul_f32 %s, %s, %s" % (B.o(2 + q), B.V(12 + q, 2), B.o(8)))
            else:
                L.append("v_pk_fma_f32 %s, %s, %s, %s" % (B.o(2 + q), B.V(12 + q, 4 * k + 2), B.o(8 + k), B.o(2 + q)))
    return B.wrap(L)


def rc2(l, acc_rows):  # cl 0-3, ch 4-7 (in/out) | a2 8, 9, bases 10, 11
    B = Block(8, 4, l)
    L = []
    for q in range(2):
        for k in range(4):
            if acc_rows:
                t = 56 + 4 * (k & 1)
                for e in range(4):
                    L.append("v_accvgpr_read_b32 v%d, a[%s+%d]" % (t + e, B.o(10 + q), 4 * k + e))
                L.append("v_pk_fma_f32 %s, v[%d:%d], %s, %s" % (B.o(k), t, t + 1, B.o(8 + q), B.o(k)))
                L.append("v_pk_fma_f32 %s, v[%d:%d], %s, %s" % (B.o(4 + k), t + 2, t + 3, B.o(8 + q), B.o(4 + k)))
            else:
                L.append("v_pk_fma_f32 %s, %s, %s, %s" % (B.o(k), B.V(10 + q, 4 * k), B.o(8 + q), B.o(k)))
                L.append("v_pk_fma_f32 %s, %s, %s, %s" % (B.o(4 + k), B.V(10 + q, 4 * k + 2), B.o(8 + q), B.o(4 + k)))
    return B.wrap(L)


T_OUT = ', "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])'
T_PAR = ", f32x4 (&t)[4], unsigned lds_addr"
CLCH = '"+v"(cl[0]), "+v"(cl[1]), "+v"(cl[2]), "+v"(cl[3]), "+v"(ch[0]), "+v"(ch[1]), "+v"(ch[2]), "+v"(ch[3])'
B8 = '"v"(blo[0]), "v"(blo[1]), "v"(blo[2]), "v"(blo[3]), "v"(bhi[0]), "v"(bhi[1]), "v"(bhi[2]), "v"(bhi[3])'


def fn(name, tparams, params, body, outs, ins, l):
    tp = tparams + (", int OFF" if l else "")
    t_out = ""
    if l:
        t_out = (", " if outs else "") + T_OUT[2:]
    return """template <%s>
__device__ __forceinline__ void %s%s(%s%s) {
    asm volatile(
%s
        : %s%s
        : %s%s);
}
""" % (tp, name, "_l" if l else "", params, T_PAR if l else "", body, outs, t_out, ins, ', "v"(lds_addr), "n"(OFF)' if l else "")


out = """// ---- the row pass of sinkhorn_resident128 / sinkhorn_resident2k on registers the compiler does not allocate ----------------------
// GENERATED by tools/gen_sk128_asm.py - edit the generator.  A K row in vector registers: v[B + 4 k + e], in accumulation
// registers: a[B + 4 k + e] (k = 256-column chunk, e = element of the lane's 4 adjacent columns); temporaries v56 - v63.  Dependent
// packed operations are never adjacent.  The `_l` form of a block issues the four ds_read_b128 of one LDS row (half) in front of its
// multiply-adds (t[c] = 16 bytes at lds_addr + OFF + 1024 c) and waits for them behind: the LDS latency hides under the block.
"""
for l in (False, True):
    out += fn("sk128_rs4v", "int B0, int B1, int B2, int B3", "f32x2 (&acc)[4], const f32x2 (&blo)[4], const f32x2 (&bhi)[4]", rs4v(l),
              '"=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3])', B8 + ', "n"(B0), "n"(B1), "n"(B2), "n"(B3)', l)
    out += fn("sk128_rs2a", "int B0, int B1", "f32x2& acc0, f32x2& acc1, const f32x2 (&blo)[4], const f32x2 (&bhi)[4]", rs2a(l),
              '"=&v"(acc0), "=&v"(acc1)', B8 + ', "n"(B0), "n"(B1)', l)
    out += fn("sk_rs2v", "int B0, int B1", "f32x2 (&acc)[4], const f32x2 (&blo)[4], const f32x2 (&bhi)[4]", rs2v(l),
              '"=&v"(acc[0]), "=&v"(acc[1]), "=&v"(acc[2]), "=&v"(acc[3])', B8 + ', "n"(B0), "n"(B1)', l)
    for nm, acc_rows in (("sk_rc2v", False), ("sk_rc2a", True)):
        out += fn(nm, "int B0, int B1", "f32x2 (&cl)[4], f32x2 (&ch)[4], const f32x2& a0, const f32x2& a1", rc2(l, acc_rows),
                  CLCH, '"s"(a0), "s"(a1), "n"(B0), "n"(B1)', l)
for nm, acc_rows in (("sk128_rc4v", False), ("sk128_rc4a", True)):
    out += fn(nm, "int B0, int B1, int B2, int B3", "f32x2 (&cl)[4], f32x2 (&ch)[4], const f32x2 (&a2)[4]", rc4(False, acc_rows),
              CLCH, '"s"(a2[0]), "s"(a2[1]), "s"(a2[2]), "s"(a2[3]), "n"(B0), "n"(B1), "n"(B2), "n"(B3)', False)

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "e2e_multi_view_matching_amd", "csrc", "sinkhorn128_rows.h")
if __name__ == "__main__":
    import sys
    if "--check" in sys.argv:  # tests/test_host_and_abi.py: the committed header IS this generator's output
        sys.exit(0 if open(path).read() == out else 1)
    open(path, "w").write(out)
    print(len(out.split("\n")), "lines ->", os.path.normpath(path))
