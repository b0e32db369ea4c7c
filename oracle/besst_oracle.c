/*
 * TEST INFRASTRUCTURE - C oracle for the integer hot loops of the scaffold-graph path.
 *
 * A sequential, single-threaded restatement of the reference's per-record logic, fast enough to check
 * the HIP path at BASELINE.json's full sizes.  It may be linked/called only by tests/, by
 * __graft_entry__.smoke() and by the cpu_baseline leg of bench.py - never by anything under besst_amd/.
 * It is validated against the Python oracle (oracle/py_oracle.py), which is itself pinned to golden
 * vectors captured from the real reference (tests/test_oracle_golden.py, tests/test_c_oracle.py).
 *
 *   oracle_record_loop     CreateGraph.PE record loop            BESST/CreateGraph.py:111-211
 *                          CreateEdge                            :812-871
 *                          PosDirCalculatorPE / MP, CheckDir     :1024-1076, :678-688
 *   oracle_record_loop_mt  the same loop on T host threads (contiguous slices; only used as the "all host cores"
 *                          CPU baseline of bench.py and checked against the sequential loop in tests/)
 *   oracle_metrics_sample  libmetrics scans                      BESST/libmetrics.py:63-84, :293-303
 *                          is_proper_aligned_unique_innie/outie  BESST/bam_parser.py:22-29
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -shared -fPIC)
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define F_UNMAPPED 0x4
#define F_MATE_UNMAPPED 0x8
#define F_REVERSE 0x10
#define F_MATE_REVERSE 0x20
#define F_READ1 0x40
#define F_READ2 0x80
#define F_SECONDARY 0x100

typedef struct {
    double read_len;
    double ins_size_threshold;
    int32_t min_mapq;
    int32_t rf;             /* orientation == 'rf' */
    int32_t detect_duplicate;
    int32_t extend_paths;
    int32_t no_score;
    int32_t node_bits;
} oracle_params;

/* counters[]: 0 count, 1 non_unique, 2 non_unique_for_scaf, 3 nr_of_duplicates, 4 too_long, 5 fishy reads,
 *             6 n_tuples, 7 n_reach, 8 prev_obs1, 9 prev_obs2 (in/out) */

/* One end of PosDirCalculatorPE/MP: int(...) truncation of a possibly fractional read_len expression. */
static void posdir(int rf, int cdir, int rdir, int64_t cpos, int64_t rpos, int64_t slen, int64_t clen, double read_len,
                   int32_t* obs, uint32_t* side) {
    if (rf) rdir = !rdir;
    if (cdir && rdir) {
        *obs = (int32_t)(slen - cpos - rpos);
        *side = 1;
    } else if (!cdir && rdir) {
        *obs = (int32_t)(cpos + (clen - rpos));
        *side = 0;
    } else if (cdir && !rdir) {
        double v = (double)(cpos + rpos) + read_len;
        *obs = (int32_t)v;
        *side = 0;
    } else {
        double inner = (double)(clen - rpos) - read_len;
        double v = (double)(slen - cpos) - inner;
        *obs = (int32_t)v;
        *side = 1;
    }
}

int64_t oracle_record_loop(int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos, const int32_t* mpos,
                           const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen, int64_t n_contigs,
                           const uint8_t* cls, const int32_t* scaf, const int32_t* slen, const int32_t* cpos,
                           const int32_t* clen, const uint8_t* cdir, const oracle_params* p, int64_t* aligned,
                           int64_t* counters, uint64_t* keys, uint64_t* payload) {
    int64_t n_tuples = 0;
    int32_t prev1 = (int32_t)counters[8], prev2 = (int32_t)counters[9];
    const int dbl_a = p->extend_paths && !p->no_score;
    const uint32_t mask_a = p->no_score ? 2u : (1u | (p->extend_paths ? 2u : 0u));
    for (int64_t i = 0; i < n; ++i) {
        const int32_t t = tid[i], m = mtid[i];
        if (t < 0 || t >= n_contigs || m < 0 || m >= n_contigs) continue;     /* KeyError -> continue (:118-124) */
        if (cls[t] == 0 || cls[m] == 0) continue;                             /* (:127-130) */
        const uint32_t f = flag[i];
        const int32_t q = mapq[i];
        if (q >= p->min_mapq || q == 0) aligned[t] += qlen[i];                /* (:138-139) */
        const int rdir = !(f & F_REVERSE), mdir = !(f & F_MATE_REVERSE);
        if ((f & F_UNMAPPED) && (f & F_READ1) && scaf[t] != scaf[m]) {        /* fishy (:141-163) */
            int32_t dummy;
            uint32_t s1, s2;
            posdir(p->rf, cdir[t], rdir, 0, 0, 0, 0, 0.0, &dummy, &s1);
            posdir(p->rf, cdir[m], mdir, 0, 0, 0, 0, 0.0, &dummy, &s2);
            const uint64_t a = (uint64_t)scaf[t] * 2 + s1, b = (uint64_t)scaf[m] * 2 + s2;
            const uint64_t lo = a < b ? a : b, hi = a < b ? b : a;
            keys[n_tuples] = (((lo << p->node_bits) | hi) << 1) | 1u;
            payload[n_tuples] = 0;
            n_tuples++;
            counters[5]++;
        }
        if (t != m && q == 0) counters[1]++;                                  /* (:166-167) */
        if (!(t != m && (f & F_READ2) && !(f & F_UNMAPPED) && q >= p->min_mapq)) continue;   /* (:169) */
        uint32_t mask;
        int calls;
        if (cls[t] == 1 && cls[m] == 1 && scaf[t] != scaf[m]) {               /* case A (:170-183) */
            mask = mask_a;
            calls = dbl_a ? 2 : 1;
        } else if (p->extend_paths) {                                         /* case B (:184-206) */
            const int st = cls[t] == 2, sm = cls[m] == 2;
            if ((st && sm && scaf[t] != scaf[m]) || (st != sm)) { mask = 2u; calls = 1; }
            else continue;
        } else {
            continue;
        }
        int32_t o1, o2;
        uint32_t s1, s2;
        posdir(p->rf, cdir[t], rdir, cpos[t], pos[i], slen[t], clen[t], p->read_len, &o1, &s1);
        posdir(p->rf, cdir[m], mdir, cpos[m], mpos[i], slen[m], clen[m], p->read_len, &o2, &s2);
        const int accept = ((double)((int64_t)o1 + o2) < p->ins_size_threshold) && o1 > 25 && o2 > 25;
        int keep = 0;
        counters[7]++;
        for (int call = 0; call < calls; ++call) {                            /* CreateEdge (:812-871) */
            if (q == 0) counters[2]++;
            const int32_t c1 = call == 0 ? prev1 : -1, c2 = call == 0 ? prev2 : -1;
            if (o1 == c1 && o2 == c2) {
                counters[3]++;
                if (p->detect_duplicate) break;
            }
            if (accept) { counters[0]++; keep = 1; } else counters[4]++;
            prev1 = o1; prev2 = o2;
        }
        if (keep) {
            const uint64_t a = (uint64_t)scaf[t] * 2 + s1, b = (uint64_t)scaf[m] * 2 + s2;
            const int first_min = a < b;
            const uint64_t lo = first_min ? a : b, hi = first_min ? b : a;
            const uint32_t plo = (uint32_t)(first_min ? o1 : o2);
            const uint32_t phi = (uint32_t)(first_min ? o2 : o1) | (mask << 30);
            keys[n_tuples] = ((lo << p->node_bits) | hi) << 1;
            payload[n_tuples] = (uint64_t)plo | ((uint64_t)phi << 32);
            n_tuples++;
        }
    }
    counters[6] += n_tuples;
    counters[8] = prev1;
    counters[9] = prev2;
    return n_tuples;
}

/* ---- multi-threaded baseline ------------------------------------------------------------------------------
 * The only coupling between records is CreateEdge's prev_obs = observation of the previous record that reached
 * CreateEdge (it is updated whether or not that record was accepted or a duplicate).  A slice therefore finds
 * its incoming prev_obs by walking BACKWARDS from its first record to the nearest reaching record (a few
 * hundred records on a real stream) and then runs the sequential loop on private outputs, merged at the end. */
static int reaches(int64_t i, const int32_t* tid, const int32_t* mtid, const int32_t* pos, const int32_t* mpos,
                   const uint16_t* flag, const uint8_t* mapq, int64_t n_contigs, const uint8_t* cls,
                   const int32_t* scaf, const int32_t* slen, const int32_t* cpos, const int32_t* clen,
                   const uint8_t* cdir, const oracle_params* p, int32_t* o1, int32_t* o2) {
    const int32_t t = tid[i], m = mtid[i];
    if (t < 0 || t >= n_contigs || m < 0 || m >= n_contigs) return 0;
    if (cls[t] == 0 || cls[m] == 0) return 0;
    const uint32_t f = flag[i];
    const int32_t q = mapq[i];
    if (!(t != m && (f & F_READ2) && !(f & F_UNMAPPED) && q >= p->min_mapq)) return 0;
    if (!(cls[t] == 1 && cls[m] == 1 && scaf[t] != scaf[m])) {
        if (!p->extend_paths) return 0;
        const int st = cls[t] == 2, sm = cls[m] == 2;
        if (!((st && sm && scaf[t] != scaf[m]) || (st != sm))) return 0;
    }
    uint32_t s1, s2;
    posdir(p->rf, cdir[t], !(f & F_REVERSE), cpos[t], pos[i], slen[t], clen[t], p->read_len, o1, &s1);
    posdir(p->rf, cdir[m], !(f & F_MATE_REVERSE), cpos[m], mpos[i], slen[m], clen[m], p->read_len, o2, &s2);
    return 1;
}

typedef struct {
    int64_t start, end, n_contigs, n_tuples;
    const int32_t *tid, *mtid, *pos, *mpos;
    const uint16_t *flag, *qlen;
    const uint8_t* mapq;
    const uint8_t *cls, *cdir;
    const int32_t *scaf, *slen, *cpos, *clen;
    const oracle_params* p;
    int64_t* aligned;      /* private, n_contigs */
    int64_t counters[10];
    uint64_t *keys, *payload;   /* shared arrays; this slice writes at [start, ...) */
} slice_job;

static void* slice_main(void* arg) {
    slice_job* j = (slice_job*)arg;
    for (int64_t i = j->start - 1; i >= 0; --i) {       /* incoming prev_obs (counters[8..9] preset to the caller's) */
        int32_t o1, o2;
        if (reaches(i, j->tid, j->mtid, j->pos, j->mpos, j->flag, j->mapq, j->n_contigs, j->cls, j->scaf, j->slen,
                    j->cpos, j->clen, j->cdir, j->p, &o1, &o2)) {
            j->counters[8] = o1;
            j->counters[9] = o2;
            break;
        }
    }
    const int64_t s = j->start;
    j->n_tuples = oracle_record_loop(j->end - s, j->tid + s, j->mtid + s, j->pos + s, j->mpos + s, j->flag + s,
                                     j->mapq + s, j->qlen + s, j->n_contigs, j->cls, j->scaf, j->slen, j->cpos,
                                     j->clen, j->cdir, j->p, j->aligned, j->counters, j->keys + s, j->payload + s);
    return NULL;
}

int64_t oracle_record_loop_mt(int n_threads, int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* pos,
                              const int32_t* mpos, const uint16_t* flag, const uint8_t* mapq, const uint16_t* qlen,
                              int64_t n_contigs, const uint8_t* cls, const int32_t* scaf, const int32_t* slen,
                              const int32_t* cpos, const int32_t* clen, const uint8_t* cdir, const oracle_params* p,
                              int64_t* aligned, int64_t* counters, uint64_t* keys, uint64_t* payload) {
    if (n_threads < 1) n_threads = 1;
    if ((int64_t)n_threads > n) n_threads = n > 0 ? (int)n : 1;
    slice_job* jobs = (slice_job*)calloc((size_t)n_threads, sizeof(slice_job));
    pthread_t* th = (pthread_t*)calloc((size_t)n_threads, sizeof(pthread_t));
    for (int t = 0; t < n_threads; ++t) {
        slice_job* j = &jobs[t];
        j->start = n * t / n_threads;
        j->end = n * (t + 1) / n_threads;
        j->n_contigs = n_contigs;
        j->tid = tid; j->mtid = mtid; j->pos = pos; j->mpos = mpos; j->flag = flag; j->mapq = mapq; j->qlen = qlen;
        j->cls = cls; j->scaf = scaf; j->slen = slen; j->cpos = cpos; j->clen = clen; j->cdir = cdir;
        j->p = p;
        j->aligned = (int64_t*)calloc((size_t)n_contigs, sizeof(int64_t));
        j->counters[8] = counters[8];
        j->counters[9] = counters[9];
        j->keys = keys;
        j->payload = payload;
        pthread_create(&th[t], NULL, slice_main, j);
    }
    int64_t n_tuples = 0;
    for (int t = 0; t < n_threads; ++t) {
        pthread_join(th[t], NULL);
        slice_job* j = &jobs[t];
        memmove(keys + n_tuples, keys + j->start, (size_t)j->n_tuples * sizeof(uint64_t));
        memmove(payload + n_tuples, payload + j->start, (size_t)j->n_tuples * sizeof(uint64_t));
        n_tuples += j->n_tuples;
        for (int64_t c = 0; c < n_contigs; ++c) aligned[c] += j->aligned[c];
        for (int f = 0; f < 8; ++f) counters[f] += j->counters[f];
        if (j->counters[7]) { counters[8] = j->counters[8]; counters[9] = j->counters[9]; }
        free(j->aligned);
    }
    free(jobs);
    free(th);
    return n_tuples;
}

static int oriented(uint32_t f, int32_t tlen, int32_t t, int32_t m, int sign) {
    const int rev = (f & F_REVERSE) != 0, mrev = (f & F_MATE_REVERSE) != 0;
    const int64_t tl = (int64_t)tlen * sign;
    return ((rev && !mrev && tl < 0) || (!rev && mrev && tl > 0)) && (f & F_READ2) && t == m;
}

static int unique_pair(uint32_t f, int32_t q, int32_t thr) {
    return !(f & F_MATE_UNMAPPED) && q > thr && !(f & F_SECONDARY);
}

/* counts[]: 0 n_isize, 1 n_contam, 2 counter_total, 3 sample_counter */
void oracle_metrics_sample(int64_t n, const int32_t* tid, const int32_t* mtid, const int32_t* tlen, const uint16_t* flag,
                           const uint8_t* mapq, int64_t n_contigs, const uint8_t* top, int rf, int32_t min_mapq,
                           double read_len, int want_isize, int32_t* isize_out, int32_t* contam_out, int64_t* counts) {
    const int64_t cap = 1000000;
    int64_t n_isize = 0, n_contam = 0, counter_total = 0, sample_counter = 0;
    int contam_done = 0;
    for (int64_t i = 0; i < n; ++i) {
        const int32_t t = tid[i];
        const int is_top = t >= 0 && t < n_contigs && top[t];
        const uint32_t f = flag[i];
        const int64_t at = tlen[i] < 0 ? -(int64_t)tlen[i] : tlen[i];
        const int innie = oriented(f, tlen[i], t, mtid[i], 1) && unique_pair(f, mapq[i], min_mapq);
        const int outie = oriented(f, tlen[i], t, mtid[i], -1) && unique_pair(f, mapq[i], min_mapq);
        if (want_isize && n_isize < cap && is_top && (rf ? outie : innie)) isize_out[n_isize++] = (int32_t)at;
        if (!contam_done && is_top) {
            sample_counter++;
            if (!(f & F_UNMAPPED)) counter_total++;
            if (!rf && outie && read_len < (double)at + 2.0 * read_len) contam_out[n_contam++] = (int32_t)at;
            if (rf && innie && read_len < (double)at) contam_out[n_contam++] = (int32_t)at;
            if (sample_counter >= cap) contam_done = 1;
        }
        if (contam_done && (!want_isize || n_isize >= cap)) break;
    }
    counts[0] = n_isize;
    counts[1] = n_contam;
    counts[2] = counter_total;
    counts[3] = sample_counter;
}
