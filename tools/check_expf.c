// tools/check_expf.c -- CPU check of the expf restatement used by the device code (svdf_device.h: glibc_expf): evaluates
// all 2^32 float inputs against the host libm in 16 fused/unfused variants (bit 3 = fused range reduction, the one that
// matches on FMA hosts).  gcc -O2 -ffp-contract=off -mfma -o check_expf tools/check_expf.c -lm -lpthread
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <pthread.h>
static const uint64_t T[32] = {
#include "check_expf_tab.inc"
};
static const double InvLn2N = 0x1.71547652b82fep+0 * 32;
static const double SHIFT = 0x1.8p+52;
static const double C0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, C1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, C2 = 0x1.62e42ff0c52d6p-1 / 32;
static inline uint32_t asu(float f){uint32_t u;memcpy(&u,&f,4);return u;}
static inline uint64_t asu64(double f){uint64_t u;memcpy(&u,&f,8);return u;}
static inline double asd(uint64_t u){double f;memcpy(&f,&u,8);return f;}
// variant bits: 1: z=C0*r+C1 fma; 2: y=C2*r+1 fma; 4: y=z*r2+y fma; 8: r = z - kd (no fma possible); 
static inline float my_expf(float x, int var) {
    double xd = (double)x;
    uint32_t abstop = (asu(x) >> 20) & 0x7ff;
    if (abstop >= (asu(88.0f) >> 20)) {
        if (asu(x) == asu(-INFINITY)) return 0.0f;
        if (abstop >= (asu(INFINITY) >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return INFINITY;
        if (x < -0x1.9fe368p6f) return 0.0f;
    }
    double z = InvLn2N * xd;
    double kd = z + SHIFT;
    uint64_t ki = asu64(kd);
    kd -= SHIFT;
    double r = (var & 8) ? fma(InvLn2N, xd, -kd) : z - kd;
    uint64_t t = T[ki % 32];
    t += ki << (52 - 5);
    double s = asd(t);
    double zz = (var & 1) ? fma(C0, r, C1) : C0 * r + C1;
    double r2 = r * r;
    double y = (var & 2) ? fma(C2, r, 1.0) : C2 * r + 1;
    y = (var & 4) ? fma(zz, r2, y) : zz * r2 + y;
    y = y * s;
    return (float)y;
}
static long mism[16][16];
static uint32_t firstbad[16];
struct job { uint32_t lo, hi; int id; };
static void *work(void *p) {
    struct job *j = (struct job *)p;
    for (uint64_t u = j->lo; u < j->hi; u++) {
        float x; uint32_t uu = (uint32_t)u; memcpy(&x, &uu, 4);
        float ref = expf(x);
        for (int v = 0; v < 16; v++) {
            float m = my_expf(x, v);
            if (asu(m) != asu(ref) && !(m != m && ref != ref)) { mism[v][j->id]++; if (!firstbad[v]) firstbad[v] = uu; }
        }
    }
    return 0;
}
int main() {
    pthread_t th[8]; struct job jb[8];
    for (int i = 0; i < 8; i++) { jb[i].lo = (uint32_t)((uint64_t)i << 29); jb[i].hi = (i == 7) ? 0xFFFFFFFFu : (uint32_t)((uint64_t)(i + 1) << 29); jb[i].id = i; pthread_create(&th[i], 0, work, &jb[i]); }
    for (int i = 0; i < 8; i++) pthread_join(th[i], 0);
    for (int v = 0; v < 16; v++) { long s = 0; for (int i = 0; i < 8; i++) s += mism[v][i]; printf("variant %d: %ld mismatches first %08x\n", v, s, firstbad[v]); }
    return 0;
}
