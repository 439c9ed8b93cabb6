// svdf_randstream.cpp -- libc rand() as a random-ACCESS stream.
//
// The reference draws its rank pairs with libc rand() (apex-tensor/apex_random.h:42-67), one call at a time, and the draws
// have to stay the same for a seeded run to see the same pairs.  glibc's default generator (random_r.c, TYPE_3) is the
// additive feedback recurrence  x[n] = x[n-3] + x[n-31]  (mod 2^32)  over a 31-word table, rand() = x[n] >> 1.  It is linear,
// so the value J steps ahead is a fixed combination of any 31 consecutive values: x[n+J] = sum_j a_j x[n+j] with
// sum_j a_j z^j = z^J mod (z^31 - z^28 - 1) over Z/2^32.  That turns the stream into chunks whose starting tables are known up
// front; the device expands all chunks in parallel (svdf_k_sample.hip), every user block samples with its own slice of the
// stream, and libc's table is left exactly where D sequential rand() calls would have left it.
//
// Reading and writing libc's table goes through initstate() / setstate() only: they return a pointer to the caller-visible
// state array (info word + table) of the generator that was active, the documented way to save and restore it.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "svdf_engine.h"

namespace svdf {

namespace {
const int DEG = 31, SEP = 3, MAX_TYPES = 5, TYPE_3 = 3;
}

// the last 31 values of the active generator, oldest first; false when the active generator is not the 31-word table
bool libc_rand_capture(LibcRand &s) {
    static char scratch[256];
    char *old = initstate(1u, scratch, sizeof(scratch));   // switches away; `old` now carries an up-to-date info word
    if (!old) return false;
    const int32_t *w = reinterpret_cast<const int32_t *>(old);
    const int info = w[0], type = info % MAX_TYPES, rear = info / MAX_TYPES;
    bool ok = type == TYPE_3 && rear >= 0 && rear < DEG;
    if (ok) {
        const int front = (rear + SEP) % DEG;   // the slot that is overwritten next holds the oldest value
        for (int j = 0; j < DEG; j++) s.x[j] = (uint32_t)w[1 + (front + j) % DEG];
    }
    setstate(old);
    s.handle = old;
    return ok;
}
// make x[0..30] (oldest first) the table of the generator captured before
void libc_rand_restore(const LibcRand &s) {
    static char scratch[256];
    char *old = initstate(1u, scratch, sizeof(scratch));
    int32_t *w = reinterpret_cast<int32_t *>(old);
    for (int j = 0; j < DEG; j++) w[1 + j] = (int32_t)s.x[j];
    w[0] = (DEG - SEP) * MAX_TYPES + TYPE_3;   // rear pointer at slot 28 -> front (oldest) at slot 0
    setstate(old);
}

// z^J mod (z^31 - z^28 - 1), coefficients mod 2^32
static void poly_mul(const uint32_t *a, const uint32_t *b, uint32_t *out) {
    uint32_t t[2 * DEG - 1];
    memset(t, 0, sizeof(t));
    for (int i = 0; i < DEG; i++) {
        if (!a[i]) continue;
        for (int j = 0; j < DEG; j++) t[i + j] += a[i] * b[j];
    }
    for (int d = 2 * DEG - 2; d >= DEG; d--) {   // z^d = z^(d-31) * (z^28 + 1)
        const uint32_t c = t[d];
        if (!c) continue;
        t[d - SEP] += c;
        t[d - DEG] += c;
    }
    memcpy(out, t, DEG * sizeof(uint32_t));
}
void libc_jump_poly(uint64_t J, uint32_t a[31]) {
    uint32_t base[DEG], acc[DEG];
    memset(base, 0, sizeof(base));
    memset(acc, 0, sizeof(acc));
    base[1] = 1;   // z
    acc[0] = 1;    // 1
    while (J) {
        if (J & 1) poly_mul(acc, base, acc);
        poly_mul(base, base, base);
        J >>= 1;
    }
    memcpy(a, acc, sizeof(acc));
}

// tables at the start of every chunk of C values: chunk c's table = the 31 values before its first new value
void libc_rand_chunk_states(const LibcRand &s0, long nchunks, long C, std::vector<uint32_t> &states) {
    states.resize((size_t)nchunks * DEG);
    if (nchunks == 0) return;
    memcpy(states.data(), s0.x, DEG * sizeof(uint32_t));
    uint32_t a[DEG];
    libc_jump_poly((uint64_t)C, a);
    uint32_t ext[2 * DEG - 1];
    for (long c = 0; c + 1 < nchunks; c++) {
        const uint32_t *cur = states.data() + (size_t)c * DEG;
        memcpy(ext, cur, DEG * sizeof(uint32_t));
        for (int t = 0; t < DEG - 1; t++) ext[DEG + t] = ext[t] + ext[t + DEG - SEP];   // x[t+31] = x[t] + x[t+28]
        uint32_t *nxt = states.data() + (size_t)(c + 1) * DEG;
        for (int m = 0; m < DEG; m++) {
            uint32_t acc = 0;
            for (int j = 0; j < DEG; j++) acc += a[j] * ext[m + j];
            nxt[m] = acc;
        }
    }
}

// the next n rand() results without consuming them (host; tests)
void libc_rand_peek(long n, int *out) {
    LibcRand s;
    if (!libc_rand_capture(s)) fail("svdfeature_amd: libc's rand() is not in its default 31-word mode");
    uint32_t buf[DEG];
    memcpy(buf, s.x, sizeof(buf));
    int f = 0;
    for (long i = 0; i < n; i++) {
        const uint32_t v = buf[f] + buf[(f + DEG - SEP) % DEG];
        buf[f] = v;
        out[i] = (int)(v >> 1);
        f = (f + 1) % DEG;
    }
}
// advance libc's generator by n draws without calling rand() n times
void libc_rand_skip(long n) {
    if (n <= 0) return;
    LibcRand s;
    if (!libc_rand_capture(s)) fail("svdfeature_amd: libc's rand() is not in its default 31-word mode");
    uint32_t a[DEG], ext[2 * DEG - 1];
    libc_jump_poly((uint64_t)n, a);
    memcpy(ext, s.x, DEG * sizeof(uint32_t));
    for (int t = 0; t < DEG - 1; t++) ext[DEG + t] = ext[t] + ext[t + DEG - SEP];
    LibcRand out = s;
    for (int m = 0; m < DEG; m++) {
        uint32_t acc = 0;
        for (int j = 0; j < DEG; j++) acc += a[j] * ext[m + j];
        out.x[m] = acc;
    }
    libc_rand_restore(out);
}

}  // namespace svdf
