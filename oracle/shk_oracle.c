/*
 * shk_oracle.c — CPU ORACLE for the sparrowhawk-asm assembly path.  TEST INFRASTRUCTURE ONLY.
 *
 *   *** PARITY UNPINNED ***  The reference implementation of this path
 *   (rust/sparrowhawk-asm, /root/reference/.gitmodules:1-4) is an empty submodule in the
 *   reference checkout, there is no Rust toolchain in the image, and the reference holds no
 *   test, fixture or golden vector for it (/root/reference/AGENTS.md:352-354).  This file is a
 *   plain, single-threaded restatement of ../SPEC.md, which is constrained by every observable
 *   contract of the reference:
 *     - surface / JSON schemas ........ www/src/workers/Assembler.ts:1-39,94-100,110,124,127
 *     - defaults, ranges, phases ...... www/src/components/pages/AssemblyPage.vue:26-32,53-59,
 *                                        80-86,313-321,430-432,451-619
 *     - histogram (500 bins) .......... www/src/components/KmerHistogram.vue:44-47,67-72
 *     - algorithm prose ............... docs/src/assembly.md:3-20
 *     - FASTQ / gz ingestion pattern .. rust/orphos-bridge/src/fastx_wasm.rs:9,53-70
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / reported baseline.  The product (sparrowhawk_amd/) never links,
 * loads or calls it.
 *
 * Style: obviously correct before fast.  K-mers are explicit multi-word integers, the solid set
 * is a sorted array searched by bisection, every graph query recomputes neighbours from
 * sequences.  `shko_count(ctx, naive=1)` rebuilds every window from scratch (O(k) per window);
 * `naive=0` rolls the two strands (used for the timed CPU baseline); tests check both agree.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <zlib.h>

#define MAXW 8
#define K_MIN 15
#define K_MAX 255
#define QUAL_OFFSET 33          /* SPEC S2 Q1 */
#define HISTO_BINS 500          /* SPEC S5, KmerHistogram.vue:45 */
#define FIT_ITERS 200           /* SPEC S6 */
#define MAX_ROUNDS 32           /* SPEC S9 */

typedef struct { uint64_t w[MAXW]; } kmer_t;     /* w[0] least significant */

typedef struct {
    char *seq; uint64_t len; uint64_t n_nodes; uint64_t kc;
} contig_t;

typedef struct { uint32_t a, ao, b, bo; } link_t;   /* contig ids 1-based, o: 0 '+', 1 '-' */

typedef struct shko_ctx {
    uint32_t k, W, min_count, min_qual;
    int do_fit, no_bubble, no_deadend;
    /* reads: one byte per base, 0..3 valid code, 4 invalid; read r spans [roff[r], roff[r+1]) */
    uint8_t *bases; uint64_t n_bases, cap_bases;
    uint64_t *roff; uint64_t n_reads, cap_reads;
    /* counting */
    uint64_t total_instances;
    uint64_t n_distinct; uint64_t *dkeys; uint32_t *dcnt;      /* sorted ascending */
    uint64_t histo[HISTO_BINS];
    uint32_t used_min_count; int fit_ok;
    uint64_t n_solid; uint64_t *skeys; uint32_t *scnt;          /* sorted ascending */
    uint8_t *alive;                                             /* per solid node */
    int rounds_run; uint64_t tips_removed, bubbles_removed;
    /* contigs */
    contig_t *contigs; uint64_t n_contigs;
    link_t *links; uint64_t n_links;
    char *fasta, *gfa1, *gfa2, *dot, *pre_json, *asm_json;
    char err[256];
} shko_ctx;

/* ------------------------------------------------------------------ k-mer arithmetic */

static kmer_t km_zero(void) { kmer_t z; memset(&z, 0, sizeof z); return z; }

static int km_cmp(const kmer_t *a, const kmer_t *b, uint32_t W) {
    for (int i = (int)W - 1; i >= 0; i--) {
        if (a->w[i] < b->w[i]) return -1;
        if (a->w[i] > b->w[i]) return 1;
    }
    return 0;
}

static uint32_t km_base(const kmer_t *x, uint32_t k, uint32_t i) { /* i-th base, 0 = first */
    uint32_t bit = 2 * (k - 1 - i);
    return (uint32_t)((x->w[bit >> 6] >> (bit & 63)) & 3);
}

static void km_set_base(kmer_t *x, uint32_t k, uint32_t i, uint32_t b) {
    uint32_t bit = 2 * (k - 1 - i);
    x->w[bit >> 6] &= ~((uint64_t)3 << (bit & 63));
    x->w[bit >> 6] |= ((uint64_t)b << (bit & 63));
}

static kmer_t km_revcomp(const kmer_t *x, uint32_t k) {
    kmer_t r = km_zero();
    for (uint32_t i = 0; i < k; i++) km_set_base(&r, k, k - 1 - i, 3 - km_base(x, k, i));
    return r;
}

/* append base b at the end, dropping the first base (the successor spelled seq[1..]+b) */
static kmer_t km_shift_in(const kmer_t *x, uint32_t k, uint32_t b) {
    kmer_t r = km_zero();
    for (uint32_t i = 0; i + 1 < k; i++) km_set_base(&r, k, i, km_base(x, k, i + 1));
    km_set_base(&r, k, k - 1, b);
    return r;
}

static kmer_t km_canonical(const kmer_t *x, uint32_t k, uint32_t W, int *orient) {
    kmer_t r = km_revcomp(x, k);
    if (km_cmp(x, &r, W) <= 0) { if (orient) *orient = 0; return *x; }
    if (orient) *orient = 1;
    return r;
}

static kmer_t km_load(const uint64_t *p, uint32_t W) {
    kmer_t x = km_zero();
    for (uint32_t i = 0; i < W; i++) x.w[i] = p[i];
    return x;
}

/* ------------------------------------------------------------------ ctx / reads */

shko_ctx *shko_new(uint32_t k, uint32_t min_count, uint32_t min_qual, int do_fit,
                   int no_bubble_collapse, int no_dead_end_removal) {
    if ((k & 1) == 0 || k < K_MIN || k > K_MAX) return NULL;
    shko_ctx *c = (shko_ctx *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->k = k; c->W = (2 * k + 63) / 64; c->min_count = min_count; c->min_qual = min_qual;
    c->do_fit = do_fit; c->no_bubble = no_bubble_collapse; c->no_deadend = no_dead_end_removal;
    c->cap_reads = 1024; c->roff = (uint64_t *)malloc((c->cap_reads + 1) * sizeof(uint64_t));
    c->roff[0] = 0;
    c->cap_bases = 1 << 16; c->bases = (uint8_t *)malloc(c->cap_bases);
    return c;
}

static void free_results(shko_ctx *c) {
    free(c->dkeys); free(c->dcnt); free(c->skeys); free(c->scnt); free(c->alive);
    c->dkeys = c->skeys = NULL; c->dcnt = c->scnt = NULL; c->alive = NULL;
    for (uint64_t i = 0; i < c->n_contigs; i++) free(c->contigs[i].seq);
    free(c->contigs); c->contigs = NULL; c->n_contigs = 0;
    free(c->links); c->links = NULL; c->n_links = 0;
    free(c->fasta); free(c->gfa1); free(c->gfa2); free(c->dot); free(c->pre_json); free(c->asm_json);
    c->fasta = c->gfa1 = c->gfa2 = c->dot = c->pre_json = c->asm_json = NULL;
}

void shko_free(shko_ctx *c) {
    if (!c) return;
    free_results(c); free(c->bases); free(c->roff); free(c);
}

const char *shko_last_error(shko_ctx *c) { return c->err; }

/* SPEC S2: a base is valid iff ACGTacgt and qual-33 >= min_qual.  qual == NULL: all pass. */
int shko_add_read(shko_ctx *c, const char *seq, const char *qual, uint64_t len) {
    if (c->n_reads == c->cap_reads) {
        c->cap_reads *= 2;
        c->roff = (uint64_t *)realloc(c->roff, (c->cap_reads + 1) * sizeof(uint64_t));
    }
    while (c->n_bases + len > c->cap_bases) {
        c->cap_bases *= 2; c->bases = (uint8_t *)realloc(c->bases, c->cap_bases);
    }
    for (uint64_t i = 0; i < len; i++) {
        uint8_t code;
        switch (seq[i]) {
            case 'A': case 'a': code = 0; break;
            case 'C': case 'c': code = 1; break;
            case 'G': case 'g': code = 2; break;
            case 'T': case 't': code = 3; break;
            default: code = 4;
        }
        if (qual && code < 4) {
            int q = (int)(unsigned char)qual[i] - QUAL_OFFSET;
            if (q < (int)c->min_qual) code = 4;
        }
        c->bases[c->n_bases + i] = code;
    }
    c->n_bases += len;
    c->n_reads++;
    c->roff[c->n_reads] = c->n_bases;
    return 0;
}

uint64_t shko_n_reads(shko_ctx *c) { return c->n_reads; }
uint64_t shko_n_bases(shko_ctx *c) { return c->n_bases; }

/* gunzip a whole (possibly multi-member) buffer.  fastx_wasm.rs:9,53-70 (MultiGzDecoder). */
static uint8_t *gunzip_all(const uint8_t *in, size_t n, size_t *out_n) {
    size_t cap = n * 4 + 1024, len = 0;
    uint8_t *out = (uint8_t *)malloc(cap);
    z_stream zs; memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, 15 + 16) != Z_OK) { free(out); return NULL; }
    zs.next_in = (Bytef *)in; zs.avail_in = (uInt)n;
    for (;;) {
        if (len == cap) { cap *= 2; out = (uint8_t *)realloc(out, cap); }
        zs.next_out = out + len; zs.avail_out = (uInt)(cap - len);
        int rc = inflate(&zs, Z_NO_FLUSH);
        len = cap - zs.avail_out;
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); free(out); return NULL; }
            continue;
        }
        if (rc != Z_OK) { inflateEnd(&zs); free(out); return NULL; }
        if (zs.avail_in == 0 && zs.avail_out != 0) { inflateEnd(&zs); free(out); return NULL; }
    }
    inflateEnd(&zs);
    *out_n = len;
    return out;
}

/* SPEC S1: strict 4-line FASTQ.  Returns 0, or -1 with err set. */
int shko_add_fastq(shko_ctx *c, const uint8_t *buf, uint64_t n) {
    uint8_t *tmp = NULL;
    if (n >= 2 && buf[0] == 0x1F && buf[1] == 0x8B) {
        size_t on = 0;
        tmp = gunzip_all(buf, n, &on);
        if (!tmp) { snprintf(c->err, sizeof c->err, "gzip decode failed"); return -1; }
        buf = tmp; n = on;
    }
    uint64_t p = 0, rec = 0;
    while (p < n) {
        if (buf[p] == '\n') { p++; continue; }                       /* blank line */
        if (buf[p] == '\r' && p + 1 < n && buf[p + 1] == '\n') { p += 2; continue; }
        const uint8_t *line[4]; uint64_t ll[4];
        for (int i = 0; i < 4; i++) {
            if (p >= n) {
                snprintf(c->err, sizeof c->err, "truncated FASTQ record %llu", (unsigned long long)rec);
                free(tmp); return -1;
            }
            uint64_t e = p;
            while (e < n && buf[e] != '\n') e++;
            if (e >= n && i < 3) {                                    /* lines 0..2 need a newline */
                snprintf(c->err, sizeof c->err, "truncated FASTQ record %llu", (unsigned long long)rec);
                free(tmp); return -1;
            }
            line[i] = buf + p; ll[i] = e - p;
            if (ll[i] > 0 && line[i][ll[i] - 1] == '\r') ll[i]--;
            p = e + 1;
        }
        if (ll[0] == 0 || line[0][0] != '@' || ll[2] == 0 || line[2][0] != '+' || ll[1] != ll[3]) {
            snprintf(c->err, sizeof c->err, "malformed FASTQ record %llu", (unsigned long long)rec);
            free(tmp); return -1;
        }
        shko_add_read(c, (const char *)line[1], (const char *)line[3], ll[1]);
        rec++;
    }
    free(tmp);
    return 0;
}

/* ------------------------------------------------------------------ counting (SPEC S3, S4) */

/* LSD radix sort of n records of W words each, 16-bit digits, over the low 2k bits. */
static void radix_sort(uint64_t *a, uint64_t n, uint32_t W, uint32_t k) {
    if (n < 2) return;
    uint64_t *b = (uint64_t *)malloc(n * W * sizeof(uint64_t));
    uint64_t *cnt = (uint64_t *)malloc(65536 * sizeof(uint64_t));
    uint32_t bits = 2 * k;
    uint64_t *src = a, *dst = b;
    for (uint32_t sh = 0; sh < bits; sh += 16) {
        memset(cnt, 0, 65536 * sizeof(uint64_t));
        uint32_t wi = sh >> 6, bo = sh & 63;
        for (uint64_t i = 0; i < n; i++) cnt[(src[i * W + wi] >> bo) & 0xFFFF]++;
        uint64_t s = 0;
        for (uint32_t d = 0; d < 65536; d++) { uint64_t t = cnt[d]; cnt[d] = s; s += t; }
        for (uint64_t i = 0; i < n; i++) {
            uint64_t d = (src[i * W + wi] >> bo) & 0xFFFF;
            uint64_t o = cnt[d]++;
            for (uint32_t j = 0; j < W; j++) dst[o * W + j] = src[i * W + j];
        }
        uint64_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * W * sizeof(uint64_t));
    free(b); free(cnt);
}

static void do_histo_fit_filter(shko_ctx *c);

int shko_count(shko_ctx *c, int naive) {
    free_results(c);
    const uint32_t k = c->k, W = c->W;
    /* upper bound on windows */
    uint64_t maxwin = 0;
    for (uint64_t r = 0; r < c->n_reads; r++) {
        uint64_t L = c->roff[r + 1] - c->roff[r];
        if (L >= k) maxwin += L - k + 1;
    }
    uint64_t *inst = (uint64_t *)malloc((maxwin ? maxwin : 1) * W * sizeof(uint64_t));
    if (!inst) { snprintf(c->err, sizeof c->err, "oom"); return -1; }
    uint64_t ni = 0;
    for (uint64_t r = 0; r < c->n_reads; r++) {
        const uint8_t *s = c->bases + c->roff[r];
        uint64_t L = c->roff[r + 1] - c->roff[r];
        if (L < k) continue;
        if (naive) {
            for (uint64_t i = 0; i + k <= L; i++) {
                int ok = 1;
                for (uint32_t j = 0; j < k; j++) if (s[i + j] > 3) { ok = 0; break; }
                if (!ok) continue;
                kmer_t f = km_zero();
                for (uint32_t j = 0; j < k; j++) km_set_base(&f, k, j, s[i + j]);
                kmer_t cn = km_canonical(&f, k, W, NULL);
                for (uint32_t j = 0; j < W; j++) inst[ni * W + j] = cn.w[j];
                ni++;
            }
        } else {
            /* rolling: fwd = fwd<<2|b ; rev = rev>>2 | (3-b)<<2(k-1) */
            kmer_t f = km_zero(), rv = km_zero();
            uint32_t run = 0;
            const uint32_t topbit = 2 * (k - 1);
            const uint32_t usedbits = 2 * k;
            for (uint64_t i = 0; i < L; i++) {
                uint8_t b = s[i];
                if (b > 3) { run = 0; f = km_zero(); rv = km_zero(); continue; }
                for (int j = (int)W - 1; j > 0; j--) f.w[j] = (f.w[j] << 2) | (f.w[j - 1] >> 62);
                f.w[0] = (f.w[0] << 2) | b;
                if (usedbits & 63) f.w[W - 1] &= (((uint64_t)1 << (usedbits & 63)) - 1);
                for (uint32_t j = 0; j + 1 < W; j++) rv.w[j] = (rv.w[j] >> 2) | (rv.w[j + 1] << 62);
                rv.w[W - 1] >>= 2;
                rv.w[topbit >> 6] |= (uint64_t)(3 - b) << (topbit & 63);
                if (++run >= k) {
                    const kmer_t *cn = km_cmp(&f, &rv, W) <= 0 ? &f : &rv;
                    for (uint32_t j = 0; j < W; j++) inst[ni * W + j] = cn->w[j];
                    ni++;
                }
            }
        }
    }
    c->total_instances = ni;
    radix_sort(inst, ni, W, k);
    /* run-length encode */
    uint64_t nd = 0;
    for (uint64_t i = 0; i < ni; i++)
        if (i == 0 || memcmp(inst + i * W, inst + (i - 1) * W, W * 8) != 0) nd++;
    c->dkeys = (uint64_t *)malloc((nd ? nd : 1) * W * sizeof(uint64_t));
    c->dcnt = (uint32_t *)malloc((nd ? nd : 1) * sizeof(uint32_t));
    uint64_t d = 0;
    for (uint64_t i = 0; i < ni;) {
        uint64_t j = i + 1;
        while (j < ni && memcmp(inst + j * W, inst + i * W, W * 8) == 0) j++;
        memcpy(c->dkeys + d * W, inst + i * W, W * 8);
        uint64_t n = j - i;
        c->dcnt[d] = n > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)n;     /* SPEC S4 saturation */
        d++; i = j;
    }
    c->n_distinct = nd;
    free(inst);
    do_histo_fit_filter(c);
    return 0;
}

/* SPEC S6.  h[1..500] in hist[0..499]. Returns 1 and *out on success, 0 if the fit fails. */
static int spectrum_fit(const uint64_t *hist, uint32_t *out) {
    double tot = 0.0, num2 = 0.0, den2 = 0.0;
    for (int c = 1; c <= HISTO_BINS; c++) {
        double h = (double)hist[c - 1];
        tot += h;
        if (c >= 2) { num2 += h * (double)c; den2 += h; }
    }
    if (tot == 0.0) return 0;
    double w = 0.5;
    double lam = den2 > 0.0 ? num2 / den2 : 2.0;
    if (lam < 2.0) lam = 2.0;
    for (int it = 0; it < FIT_ITERS; it++) {
        double sw = 0.0, sn = 0.0, sd = 0.0;
        double l1w = log(1.0 - w), lw = log(w), llam = log(lam);
        for (int c = 1; c <= HISTO_BINS; c++) {
            double h = (double)hist[c - 1];
            if (h == 0.0) continue;
            double lg = lgamma((double)c + 1.0);
            double lpc = (double)c * llam - lam - lg;
            double lpe = (double)c * 0.0 - 1.0 - lg;         /* log Pois(c; 1) */
            double r = 1.0 / (1.0 + exp((l1w + lpc) - (lw + lpe)));
            sw += h * r;
            sn += h * (1.0 - r) * (double)c;
            sd += h * (1.0 - r);
        }
        w = sw / tot;
        if (w < 1e-9) w = 1e-9;
        if (w > 1.0 - 1e-9) w = 1.0 - 1e-9;
        if (sd > 0.0) lam = sn / sd;
        if (lam < 1.000001) lam = 1.000001;
    }
    if (lam < 2.5) return 0;
    double l1w = log(1.0 - w), lw = log(w), llam = log(lam);
    for (int c = 2; c <= HISTO_BINS; c++) {
        double lg = lgamma((double)c + 1.0);
        double lpc = (double)c * llam - lam - lg;
        double lpe = (double)c * 0.0 - 1.0 - lg;
        if (l1w + lpc > lw + lpe) {
            int v = c - 1;
            if (v < 1) v = 1;
            if (v > 30) v = 30;
            *out = (uint32_t)v;
            return 1;
        }
    }
    return 0;
}

/* exposed so tests can fit arbitrary histograms */
int shko_fit(const uint64_t *hist500, uint32_t *out) { return spectrum_fit(hist500, out); }

static void do_histo_fit_filter(shko_ctx *c) {
    const uint32_t W = c->W;
    memset(c->histo, 0, sizeof c->histo);
    for (uint64_t i = 0; i < c->n_distinct; i++) {             /* SPEC S5 */
        uint32_t n = c->dcnt[i];
        uint32_t bin = n >= HISTO_BINS ? HISTO_BINS - 1 : n - 1;
        c->histo[bin]++;
    }
    c->used_min_count = c->min_count; c->fit_ok = 0;
    if (c->do_fit) {
        uint32_t v;
        if (spectrum_fit(c->histo, &v)) { c->used_min_count = v; c->fit_ok = 1; }
    }
    uint64_t ns = 0;
    for (uint64_t i = 0; i < c->n_distinct; i++) if (c->dcnt[i] > c->used_min_count) ns++;  /* S7 */
    c->skeys = (uint64_t *)malloc((ns ? ns : 1) * W * sizeof(uint64_t));
    c->scnt = (uint32_t *)malloc((ns ? ns : 1) * sizeof(uint32_t));
    c->alive = (uint8_t *)malloc(ns ? ns : 1);
    uint64_t s = 0;
    for (uint64_t i = 0; i < c->n_distinct; i++) if (c->dcnt[i] > c->used_min_count) {
        memcpy(c->skeys + s * W, c->dkeys + i * W, W * 8);
        c->scnt[s] = c->dcnt[i]; c->alive[s] = 1; s++;
    }
    c->n_solid = ns;
}

uint64_t shko_total_instances(shko_ctx *c) { return c->total_instances; }
uint64_t shko_n_distinct(shko_ctx *c) { return c->n_distinct; }
void shko_get_distinct(shko_ctx *c, uint64_t *keys, uint32_t *counts) {
    memcpy(keys, c->dkeys, c->n_distinct * c->W * 8);
    memcpy(counts, c->dcnt, c->n_distinct * 4);
}
void shko_get_histo(shko_ctx *c, uint64_t *out) { memcpy(out, c->histo, sizeof c->histo); }
uint32_t shko_used_min_count(shko_ctx *c) { return c->used_min_count; }
int shko_fit_ok(shko_ctx *c) { return c->fit_ok; }
uint64_t shko_n_solid(shko_ctx *c) { return c->n_solid; }
void shko_get_solid(shko_ctx *c, uint64_t *keys, uint32_t *counts) {
    memcpy(keys, c->skeys, c->n_solid * c->W * 8);
    memcpy(counts, c->scnt, c->n_solid * 4);
}

/* ------------------------------------------------------------------ graph (SPEC S8) */

typedef int64_t onode_t;      /* oriented node: idx*2 + o ; -1 = none */

static int64_t find_solid(const shko_ctx *c, const kmer_t *x) {
    int64_t lo = 0, hi = (int64_t)c->n_solid - 1;
    while (lo <= hi) {
        int64_t mid = lo + (hi - lo) / 2;
        kmer_t m = km_load(c->skeys + (uint64_t)mid * c->W, c->W);
        int r = km_cmp(&m, x, c->W);
        if (r == 0) return mid;
        if (r < 0) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

static kmer_t node_seq(const shko_ctx *c, onode_t v) {
    kmer_t x = km_load(c->skeys + (uint64_t)(v >> 1) * c->W, c->W);
    return (v & 1) ? km_revcomp(&x, c->k) : x;
}

static onode_t onode_rc(onode_t v) { return v ^ 1; }

/* out-neighbour of v by appended base b among ALIVE nodes, or -1 */
static onode_t out_nbr(const shko_ctx *c, onode_t v, uint32_t b) {
    kmer_t s = node_seq(c, v);
    kmer_t t = km_shift_in(&s, c->k, b);
    int o;
    kmer_t cn = km_canonical(&t, c->k, c->W, &o);
    int64_t idx = find_solid(c, &cn);
    if (idx < 0 || !c->alive[idx]) return -1;
    return idx * 2 + o;
}

static int out_nbrs(const shko_ctx *c, onode_t v, onode_t out[4]) {
    int n = 0;
    for (uint32_t b = 0; b < 4; b++) { onode_t u = out_nbr(c, v, b); if (u >= 0) out[n++] = u; }
    return n;
}
static int outdeg(const shko_ctx *c, onode_t v) { onode_t t[4]; return out_nbrs(c, v, t); }
static int indeg(const shko_ctx *c, onode_t v) { return outdeg(c, onode_rc(v)); }

/* adjacency byte of node idx (SPEC S8) over alive nodes */
static uint8_t adjacency(const shko_ctx *c, uint64_t idx) {
    uint8_t a = 0;
    for (uint32_t b = 0; b < 4; b++) {
        if (out_nbr(c, (onode_t)idx * 2, b) >= 0) a |= (uint8_t)(1u << b);
        /* predecessor spelled b + seq[..k-1): it is rc of the successor of rc(x) by base 3-b */
        if (out_nbr(c, (onode_t)idx * 2 + 1, 3 - b) >= 0) a |= (uint8_t)(1u << (4 + b));
    }
    return a;
}

void shko_get_adjacency(shko_ctx *c, uint8_t *adj) {
    for (uint64_t i = 0; i < c->n_solid; i++) adj[i] = c->alive[i] ? adjacency(c, i) : 0;
}
void shko_get_alive(shko_ctx *c, uint8_t *alive) { memcpy(alive, c->alive, c->n_solid); }

/* ------------------------------------------------------------------ correction (SPEC S9) */

typedef struct { onode_t junction; uint64_t len; uint64_t sum; kmer_t first; uint64_t start_idx;
                 onode_t *path; } tip_t;

static int tip_better(const shko_ctx *c, const tip_t *a, const tip_t *b) { /* a better than b */
    if (a->len != b->len) return a->len > b->len;
    if (a->sum != b->sum) return a->sum > b->sum;
    return km_cmp(&a->first, &b->first, c->W) < 0;
}

static const tip_t *g_sort_tips;
static int tip_ord_cmp(const void *pa, const void *pb) {
    uint64_t a = *(const uint64_t *)pa, b = *(const uint64_t *)pb;
    onode_t ja = g_sort_tips[a].junction, jb = g_sort_tips[b].junction;
    if (ja != jb) return ja < jb ? -1 : 1;
    return a < b ? -1 : (a > b ? 1 : 0);
}

static uint64_t tip_round(shko_ctx *c) {
    const uint64_t T_TIP = 2ull * c->k;
    uint64_t ntips = 0, cap = 64;
    tip_t *tips = (tip_t *)malloc(cap * sizeof(tip_t));
    for (uint64_t i = 0; i < c->n_solid; i++) {
        if (!c->alive[i]) continue;
        for (int o = 0; o < 2; o++) {
            onode_t v = (onode_t)i * 2 + o;
            if (indeg(c, v) != 0) continue;
            onode_t *path = (onode_t *)malloc((T_TIP + 1) * sizeof(onode_t));
            uint64_t plen = 0; path[plen++] = v;
            onode_t cur = v; onode_t J = -1;
            for (;;) {
                onode_t nb[4];
                if (out_nbrs(c, cur, nb) != 1) break;
                onode_t n = nb[0];
                if (indeg(c, n) >= 2) { J = n; break; }
                path[plen++] = n; cur = n;
                if (plen > T_TIP) break;
            }
            if (J < 0 || plen > T_TIP) { free(path); continue; }
            if (ntips == cap) { cap *= 2; tips = (tip_t *)realloc(tips, cap * sizeof(tip_t)); }
            tip_t *t = &tips[ntips++];
            t->junction = J; t->len = plen; t->sum = 0; t->path = path;
            for (uint64_t j = 0; j < plen; j++) t->sum += c->scnt[path[j] >> 1];
            t->first = km_load(c->skeys + (uint64_t)(v >> 1) * c->W, c->W);
        }
    }
    /* decide per junction on the snapshot: group tips by junction */
    uint8_t *kill = (uint8_t *)calloc(ntips ? ntips : 1, 1);
    uint64_t *ord = (uint64_t *)malloc((ntips ? ntips : 1) * sizeof(uint64_t));
    for (uint64_t a = 0; a < ntips; a++) ord[a] = a;
    g_sort_tips = tips;
    qsort(ord, ntips, sizeof(uint64_t), tip_ord_cmp);
    for (uint64_t g0 = 0; g0 < ntips;) {
        uint64_t g1 = g0;
        while (g1 < ntips && tips[ord[g1]].junction == tips[ord[g0]].junction) g1++;
        uint64_t t = g1 - g0;
        uint64_t d = (uint64_t)indeg(c, tips[ord[g0]].junction);
        if (t < d) {
            for (uint64_t q = g0; q < g1; q++) kill[ord[q]] = 1;
        } else {                                   /* every in-branch is a tip: keep the best */
            uint64_t best = ord[g0];
            for (uint64_t q = g0 + 1; q < g1; q++)
                if (tip_better(c, &tips[ord[q]], &tips[best])) best = ord[q];
            for (uint64_t q = g0; q < g1; q++) kill[ord[q]] = (ord[q] != best);
        }
        g0 = g1;
    }
    free(ord);
    uint64_t removed = 0;
    for (uint64_t a = 0; a < ntips; a++) {
        if (kill[a]) for (uint64_t j = 0; j < tips[a].len; j++) {
            uint64_t idx = (uint64_t)(tips[a].path[j] >> 1);
            if (c->alive[idx]) { c->alive[idx] = 0; removed++; }
        }
        free(tips[a].path);
    }
    free(kill); free(tips);
    return removed;
}

typedef struct { onode_t end; uint64_t len; uint64_t sum; kmer_t first; onode_t path[2 * K_MAX + 2]; int ok; } branch_t;

static int onode_key_le(const shko_ctx *c, onode_t a, onode_t b) {  /* key(x,o) = (x,o) */
    kmer_t ka = km_load(c->skeys + (uint64_t)(a >> 1) * c->W, c->W);
    kmer_t kb = km_load(c->skeys + (uint64_t)(b >> 1) * c->W, c->W);
    int r = km_cmp(&ka, &kb, c->W);
    if (r != 0) return r < 0;
    return (a & 1) <= (b & 1);
}

static int branch_better(const shko_ctx *c, const branch_t *a, const branch_t *b) {
    uint64_t l = a->sum * b->len, r = b->sum * a->len;
    if (l != r) return l > r;
    if (a->len != b->len) return a->len < b->len;
    return km_cmp(&a->first, &b->first, c->W) < 0;
}

static uint64_t bubble_round(shko_ctx *c) {
    const uint64_t T_BUB = 2ull * c->k;
    uint8_t *mark = (uint8_t *)calloc(c->n_solid ? c->n_solid : 1, 1);
    for (uint64_t i = 0; i < c->n_solid; i++) {
        if (!c->alive[i]) continue;
        for (int o = 0; o < 2; o++) {
            onode_t S = (onode_t)i * 2 + o;
            onode_t nb[4];
            int d = out_nbrs(c, S, nb);
            if (d < 2) continue;
            branch_t br[4];
            for (int j = 0; j < d; j++) {
                branch_t *B = &br[j];
                B->ok = 0; B->len = 0; B->sum = 0; B->end = -1;
                onode_t b = nb[j];
                if (indeg(c, b) != 1) continue;
                B->path[B->len++] = b;
                onode_t cur = b;
                for (;;) {
                    onode_t nn[4];
                    if (out_nbrs(c, cur, nn) != 1) break;
                    onode_t n = nn[0];
                    int id = indeg(c, n);
                    if (id >= 2) { B->end = n; B->ok = 1; break; }
                    if (B->len + 1 > T_BUB) break;
                    B->path[B->len++] = n; cur = n;
                }
                if (!B->ok) continue;
                for (uint64_t q = 0; q < B->len; q++) B->sum += c->scnt[B->path[q] >> 1];
                B->first = km_load(c->skeys + (uint64_t)(b >> 1) * c->W, c->W);
            }
            for (int a = 0; a < d; a++) {
                if (!br[a].ok) continue;
                onode_t E = br[a].end;
                if (!onode_key_le(c, S, onode_rc(E))) continue;     /* evaluate from one side */
                int grp = 0, best = 1;
                for (int b2 = 0; b2 < d; b2++) {
                    if (!br[b2].ok || br[b2].end != E) continue;
                    grp++;
                    if (b2 != a && branch_better(c, &br[b2], &br[a])) best = 0;
                }
                if (grp >= 2 && !best)
                    for (uint64_t q = 0; q < br[a].len; q++) mark[br[a].path[q] >> 1] = 1;
            }
        }
    }
    uint64_t removed = 0;
    for (uint64_t i = 0; i < c->n_solid; i++) if (mark[i] && c->alive[i]) { c->alive[i] = 0; removed++; }
    free(mark);
    return removed;
}

int shko_correct(shko_ctx *c) {
    c->rounds_run = 0; c->tips_removed = c->bubbles_removed = 0;
    for (int r = 0; r < MAX_ROUNDS; r++) {
        uint64_t n1 = c->no_deadend ? 0 : tip_round(c);
        uint64_t n2 = c->no_bubble ? 0 : bubble_round(c);
        c->tips_removed += n1; c->bubbles_removed += n2; c->rounds_run++;
        if (n1 + n2 == 0) break;
    }
    return 0;
}
uint64_t shko_tips_removed(shko_ctx *c) { return c->tips_removed; }
uint64_t shko_bubbles_removed(shko_ctx *c) { return c->bubbles_removed; }

/* ------------------------------------------------------------------ collapse (SPEC S10) */

static onode_t succ_simple(const shko_ctx *c, onode_t u) {
    onode_t nb[4];
    if (out_nbrs(c, u, nb) != 1) return -1;
    onode_t v = nb[0];
    if (indeg(c, v) != 1) return -1;
    if (v == u || v == onode_rc(u)) return -1;
    return v;
}
static int has_pred(const shko_ctx *c, onode_t v) { return succ_simple(c, onode_rc(v)) >= 0; }

static const char BASES[4] = {'A', 'C', 'G', 'T'};

static void revcomp_str(char *s, uint64_t n) {
    for (uint64_t i = 0; i < n / 2; i++) { char t = s[i]; s[i] = s[n - 1 - i]; s[n - 1 - i] = t; }
    for (uint64_t i = 0; i < n; i++) {
        switch (s[i]) { case 'A': s[i] = 'T'; break; case 'C': s[i] = 'G'; break;
                        case 'G': s[i] = 'C'; break; default: s[i] = 'A'; }
    }
}

static void emit_chain(shko_ctx *c, onode_t head, uint8_t *visited, uint64_t *cap) {
    const uint32_t k = c->k;
    /* first pass: length */
    uint64_t n = 0; onode_t v = head;
    do { n++; v = succ_simple(c, v); } while (v >= 0 && v != head);
    contig_t ct; ct.n_nodes = n; ct.len = n + k - 1; ct.kc = 0;
    ct.seq = (char *)malloc(ct.len + 1);
    kmer_t s = node_seq(c, head);
    for (uint32_t i = 0; i < k; i++) ct.seq[i] = BASES[km_base(&s, k, i)];
    v = head; uint64_t pos = k;
    for (uint64_t i = 0; i < n; i++) {
        visited[v >> 1] = 1;
        ct.kc += c->scnt[v >> 1];
        if (i > 0) { kmer_t t = node_seq(c, v); ct.seq[pos++] = BASES[km_base(&t, k, k - 1)]; }
        v = succ_simple(c, v);
    }
    ct.seq[ct.len] = 0;
    /* canonical orientation: min(seq, revcomp) */
    char *r = (char *)malloc(ct.len + 1);
    memcpy(r, ct.seq, ct.len + 1); revcomp_str(r, ct.len);
    if (strcmp(r, ct.seq) < 0) { free(ct.seq); ct.seq = r; } else free(r);
    if (c->n_contigs == *cap) { *cap *= 2; c->contigs = (contig_t *)realloc(c->contigs, *cap * sizeof(contig_t)); }
    c->contigs[c->n_contigs++] = ct;
}

static int contig_cmp(const void *pa, const void *pb) {
    const contig_t *a = (const contig_t *)pa, *b = (const contig_t *)pb;
    if (a->len != b->len) return a->len > b->len ? -1 : 1;
    return strcmp(a->seq, b->seq);
}

static int link_cmp(const void *pa, const void *pb) {
    const link_t *a = (const link_t *)pa, *b = (const link_t *)pb;
    if (a->a != b->a) return a->a < b->a ? -1 : 1;
    if (a->ao != b->ao) return a->ao < b->ao ? -1 : 1;
    if (a->b != b->b) return a->b < b->b ? -1 : 1;
    if (a->bo != b->bo) return a->bo < b->bo ? -1 : 1;
    return 0;
}

/* string builder */
typedef struct { char *p; size_t n, cap; } sb_t;
static void sb_init(sb_t *s) { s->cap = 1 << 12; s->p = (char *)malloc(s->cap); s->n = 0; s->p[0] = 0; }
static void sb_add(sb_t *s, const char *t, size_t n) {
    while (s->n + n + 1 > s->cap) { s->cap *= 2; s->p = (char *)realloc(s->p, s->cap); }
    memcpy(s->p + s->n, t, n); s->n += n; s->p[s->n] = 0;
}
static void sb_str(sb_t *s, const char *t) { sb_add(s, t, strlen(t)); }
static void sb_u64(sb_t *s, uint64_t v) { char b[32]; snprintf(b, sizeof b, "%llu", (unsigned long long)v); sb_str(s, b); }
static void sb_json_str(sb_t *s, const char *t) {
    sb_str(s, "\"");
    for (; *t; t++) {
        switch (*t) {
            case '\n': sb_str(s, "\\n"); break;
            case '\t': sb_str(s, "\\t"); break;
            case '"': sb_str(s, "\\\""); break;
            case '\\': sb_str(s, "\\\\"); break;
            default: sb_add(s, t, 1);
        }
    }
    sb_str(s, "\"");
}

static kmer_t kmer_from_str(const char *p, uint32_t k) {
    kmer_t x = km_zero();
    for (uint32_t i = 0; i < k; i++) {
        uint32_t b = p[i] == 'A' ? 0 : p[i] == 'C' ? 1 : p[i] == 'G' ? 2 : 3;
        km_set_base(&x, k, i, b);
    }
    return x;
}

static onode_t onode_of_seq(const shko_ctx *c, const kmer_t *s) {
    int o; kmer_t cn = km_canonical(s, c->k, c->W, &o);
    int64_t idx = find_solid(c, &cn);
    if (idx < 0 || !c->alive[idx]) return -1;
    return idx * 2 + o;
}

static void build_outputs(shko_ctx *c);

int shko_collapse(shko_ctx *c) {
    for (uint64_t i = 0; i < c->n_contigs; i++) free(c->contigs[i].seq);
    free(c->contigs);
    uint64_t cap = 64;
    c->contigs = (contig_t *)malloc(cap * sizeof(contig_t)); c->n_contigs = 0;
    uint8_t *visited = (uint8_t *)calloc(c->n_solid ? c->n_solid : 1, 1);
    for (uint64_t i = 0; i < c->n_solid; i++) {
        if (!c->alive[i]) continue;
        for (int o = 0; o < 2; o++) {
            onode_t v = (onode_t)i * 2 + o;
            if (visited[i]) break;
            if (has_pred(c, v)) continue;
            emit_chain(c, v, visited, &cap);
        }
    }
    /* what is left lies on circular unitigs; ascending index = ascending key, so the first
       unvisited node met is the smallest key of its cycle, orientation 0 (SPEC S10). */
    for (uint64_t i = 0; i < c->n_solid; i++) {
        if (!c->alive[i] || visited[i]) continue;
        emit_chain(c, (onode_t)i * 2, visited, &cap);
    }
    free(visited);
    qsort(c->contigs, c->n_contigs, sizeof(contig_t), contig_cmp);
    build_outputs(c);
    return 0;
}

/* SPEC S11 */
static void build_outputs(shko_ctx *c) {
    const uint32_t k = c->k;
    /* links: head map by brute force over contigs (obviously correct, O(C^2) avoided by sorting) */
    uint64_t nc = c->n_contigs;
    onode_t *first_p = (onode_t *)malloc((nc ? nc : 1) * sizeof(onode_t));
    onode_t *last_p = (onode_t *)malloc((nc ? nc : 1) * sizeof(onode_t));
    for (uint64_t i = 0; i < nc; i++) {
        kmer_t f = kmer_from_str(c->contigs[i].seq, k);
        kmer_t l = kmer_from_str(c->contigs[i].seq + c->contigs[i].len - k, k);
        first_p[i] = onode_of_seq(c, &f); last_p[i] = onode_of_seq(c, &l);
    }
    /* head_of[onode] = contig*2+o + 1 (0 = none) */
    uint64_t *head_of = (uint64_t *)calloc(2 * (c->n_solid ? c->n_solid : 1), sizeof(uint64_t));
    for (uint64_t i = 0; i < nc; i++) {
        head_of[first_p[i]] = i * 2 + 0 + 1;
        if (head_of[onode_rc(last_p[i])] == 0) head_of[onode_rc(last_p[i])] = i * 2 + 1 + 1;
    }
    free(c->links); c->n_links = 0; uint64_t lcap = 64;
    c->links = (link_t *)malloc(lcap * sizeof(link_t));
    for (uint64_t i = 0; i < nc; i++) for (uint32_t o = 0; o < 2; o++) {
        onode_t tail = o == 0 ? last_p[i] : onode_rc(first_p[i]);
        onode_t nb[4]; int d = out_nbrs(c, tail, nb);
        for (int j = 0; j < d; j++) {
            /* head_of: '+' for a contig's first oriented node, '-' for rc(last) */
            uint64_t h = head_of[nb[j]];
            if (!h) continue;
            uint64_t cj = (h - 1) >> 1; uint32_t oj = (uint32_t)((h - 1) & 1);
            /* a contig that is its own reverse complement start: first == rc(last) handled by
               preferring '+' above */
            link_t L = { (uint32_t)i + 1, o, (uint32_t)cj + 1, oj };
            link_t M = { (uint32_t)cj + 1, !oj, (uint32_t)i + 1, !o };
            if (link_cmp(&M, &L) < 0) L = M;
            if (c->n_links == lcap) { lcap *= 2; c->links = (link_t *)realloc(c->links, lcap * sizeof(link_t)); }
            c->links[c->n_links++] = L;
        }
    }
    qsort(c->links, c->n_links, sizeof(link_t), link_cmp);
    uint64_t u = 0;
    for (uint64_t i = 0; i < c->n_links; i++)
        if (u == 0 || link_cmp(&c->links[i], &c->links[u - 1]) != 0) c->links[u++] = c->links[i];
    c->n_links = u;
    free(first_p); free(last_p); free(head_of);

    sb_t fa, g1, g2, dt; sb_init(&fa); sb_init(&g1); sb_init(&g2); sb_init(&dt);
    sb_str(&g1, "H\tVN:Z:1.0\n"); sb_str(&g2, "H\tVN:Z:2.0\n"); sb_str(&dt, "digraph sparrowhawk {\n");
    for (uint64_t i = 0; i < nc; i++) {
        contig_t *ct = &c->contigs[i];
        sb_str(&fa, ">contig_"); sb_u64(&fa, i + 1); sb_str(&fa, " len="); sb_u64(&fa, ct->len);
        sb_str(&fa, " kc="); sb_u64(&fa, ct->kc); sb_str(&fa, "\n"); sb_add(&fa, ct->seq, ct->len); sb_str(&fa, "\n");
        sb_str(&g1, "S\t"); sb_u64(&g1, i + 1); sb_str(&g1, "\t"); sb_add(&g1, ct->seq, ct->len);
        sb_str(&g1, "\tLN:i:"); sb_u64(&g1, ct->len); sb_str(&g1, "\tKC:i:"); sb_u64(&g1, ct->kc); sb_str(&g1, "\n");
        sb_str(&g2, "S\t"); sb_u64(&g2, i + 1); sb_str(&g2, "\t"); sb_u64(&g2, ct->len); sb_str(&g2, "\t");
        sb_add(&g2, ct->seq, ct->len); sb_str(&g2, "\tKC:i:"); sb_u64(&g2, ct->kc); sb_str(&g2, "\n");
        sb_str(&dt, "  \""); sb_u64(&dt, i + 1); sb_str(&dt, "\" [label=\""); sb_u64(&dt, i + 1);
        sb_str(&dt, " len="); sb_u64(&dt, ct->len); sb_str(&dt, " kc="); sb_u64(&dt, ct->kc); sb_str(&dt, "\"];\n");
    }
    for (uint64_t i = 0; i < c->n_links; i++) {
        link_t *L = &c->links[i];
        uint64_t la = c->contigs[L->a - 1].len, lb = c->contigs[L->b - 1].len;
        sb_str(&g1, "L\t"); sb_u64(&g1, L->a); sb_str(&g1, L->ao ? "\t-\t" : "\t+\t"); sb_u64(&g1, L->b);
        sb_str(&g1, L->bo ? "\t-\t" : "\t+\t"); sb_u64(&g1, k - 1); sb_str(&g1, "M\n");
        sb_str(&g2, "E\t*\t"); sb_u64(&g2, L->a); sb_str(&g2, L->ao ? "-\t" : "+\t"); sb_u64(&g2, L->b);
        sb_str(&g2, L->bo ? "-\t" : "+\t");
        if (!L->ao) { sb_u64(&g2, la - (k - 1)); sb_str(&g2, "\t"); sb_u64(&g2, la); sb_str(&g2, "$\t"); }
        else { sb_str(&g2, "0\t"); sb_u64(&g2, k - 1); if (k - 1 == la) sb_str(&g2, "$"); sb_str(&g2, "\t"); }
        if (!L->bo) { sb_str(&g2, "0\t"); sb_u64(&g2, k - 1); if (k - 1 == lb) sb_str(&g2, "$"); sb_str(&g2, "\t"); }
        else { sb_u64(&g2, lb - (k - 1)); sb_str(&g2, "\t"); sb_u64(&g2, lb); sb_str(&g2, "$\t"); }
        sb_u64(&g2, k - 1); sb_str(&g2, "M\n");
        sb_str(&dt, "  \""); sb_u64(&dt, L->a); sb_str(&dt, "\" -> \""); sb_u64(&dt, L->b);
        sb_str(&dt, "\" [label=\""); sb_str(&dt, L->ao ? "-" : "+"); sb_str(&dt, L->bo ? "-" : "+"); sb_str(&dt, "\"];\n");
    }
    sb_str(&dt, "}\n");
    free(c->fasta); free(c->gfa1); free(c->gfa2); free(c->dot);
    c->fasta = fa.p; c->gfa1 = g1.p; c->gfa2 = g2.p; c->dot = dt.p;

    sb_t js; sb_init(&js);
    sb_str(&js, "{\"outfasta\":"); sb_json_str(&js, c->fasta);
    sb_str(&js, ",\"ncontigs\":"); sb_u64(&js, nc);
    sb_str(&js, ",\"outdot\":"); sb_json_str(&js, c->dot);
    sb_str(&js, ",\"outgfa\":"); sb_json_str(&js, c->gfa1);
    sb_str(&js, ",\"outgfav2\":"); sb_json_str(&js, c->gfa2); sb_str(&js, "}");
    free(c->asm_json); c->asm_json = js.p;
}

const char *shko_preprocessing_json(shko_ctx *c) {
    sb_t js; sb_init(&js);
    sb_str(&js, "{\"nkmers\":"); sb_u64(&js, c->n_solid); sb_str(&js, ",\"histo\":[");
    for (int i = 0; i < HISTO_BINS; i++) { if (i) sb_str(&js, ","); sb_u64(&js, c->histo[i]); }
    sb_str(&js, "],\"used_min_count\":"); sb_u64(&js, c->used_min_count); sb_str(&js, "}");
    free(c->pre_json); c->pre_json = js.p;
    return c->pre_json;
}

/* whole assemble() = graph (implicit) + correct + collapse + outputs */
int shko_assemble(shko_ctx *c) {
    if (!c->skeys) { snprintf(c->err, sizeof c->err, "count first"); return -1; }
    shko_correct(c);
    return shko_collapse(c);
}

uint64_t shko_n_contigs(shko_ctx *c) { return c->n_contigs; }
uint64_t shko_contig_len(shko_ctx *c, uint64_t i) { return c->contigs[i].len; }
uint64_t shko_contig_kc(shko_ctx *c, uint64_t i) { return c->contigs[i].kc; }
const char *shko_contig_seq(shko_ctx *c, uint64_t i) { return c->contigs[i].seq; }
const char *shko_fasta(shko_ctx *c) { return c->fasta; }
const char *shko_gfa1(shko_ctx *c) { return c->gfa1; }
const char *shko_gfa2(shko_ctx *c) { return c->gfa2; }
const char *shko_dot(shko_ctx *c) { return c->dot; }
const char *shko_assembly_json(shko_ctx *c) { return c->asm_json; }
