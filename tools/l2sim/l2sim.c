/* l2sim.c -- trace-driven model of the per-XCD L2 under the gather part of the SpMM (tools/l2sim/README.md).
 *
 * One XCD works on the tasks of its slice: a window of W tasks is "resident" (CUs x workgroups x tasks per
 * workgroup), every resident task issues its next batch of U entries per round (the kernel keeps U row loads in
 * flight per task), a finished task is replaced by the next one of the slice's list.  Every entry touches the
 * `lines` 128-byte lines of its feature row that the current feature pass covers.  The L2 is `sets` x `ways`
 * lines, LRU, indexed by a multiplicative hash of the line address (the hardware hashes the upper address bits:
 * r02 probe 4 found no set-conflict problem with col % 8 slicing).  Streams (the (col, val) pairs, the partial-sum
 * slots) are not modelled.  Returns hits / misses per slice: misses x 128 B = the fabric reads of feature rows.
 *
 * gcc -O2 -shared -fPIC -o l2sim.so l2sim.c
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t *tag; uint32_t *stamp; int sets, ways; uint32_t clock; } cache_t;

static inline int cache_access(cache_t *c, uint64_t line) {
    const uint64_t h = line * 0x9E3779B97F4A7C15ull;
    const int set = (int)((h >> 40) % (uint64_t)c->sets);
    uint64_t *t = c->tag + (size_t)set * c->ways;
    uint32_t *s = c->stamp + (size_t)set * c->ways;
    int victim = 0;
    uint32_t oldest = 0xffffffffu;
    ++c->clock;
    for (int w = 0; w < c->ways; ++w) {
        if (t[w] == line + 1) { s[w] = c->clock; return 1; }
        if (s[w] < oldest) { oldest = s[w]; victim = w; }
    }
    t[victim] = line + 1;
    s[victim] = c->clock;
    return 0;
}

/* tasks: ntasks x {kbeg, len} (int64) in execution order of ONE slice; col: column of every stored entry;
 * passes x lines_per_pass lines per row (row = passes * lines_per_pass lines of 128 B) */
int l2sim_slice(const int64_t *tasks, int64_t ntasks, const int32_t *col, int window, int batch, int passes,
                int lines_per_pass, int sets, int ways, int64_t *hits_out, int64_t *misses_out) {
    cache_t c;
    c.sets = sets; c.ways = ways; c.clock = 0;
    c.tag = (uint64_t *)calloc((size_t)sets * ways, sizeof(uint64_t));
    c.stamp = (uint32_t *)calloc((size_t)sets * ways, sizeof(uint32_t));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * window);      /* task id in each window seat (-1: empty) */
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * window);      /* entries already issued by that task */
    if (!c.tag || !c.stamp || !cur || !pos) return -1;
    int64_t hits = 0, misses = 0;
    const int row_lines = passes * lines_per_pass;
    for (int p = 0; p < passes; ++p) {
        int64_t next = 0, active = 0;
        for (int s = 0; s < window; ++s) { cur[s] = next < ntasks ? next++ : -1; pos[s] = 0; if (cur[s] >= 0) ++active; }
        while (active > 0) {
            for (int s = 0; s < window; ++s) {
                const int64_t t = cur[s];
                if (t < 0) continue;
                const int64_t kbeg = tasks[2 * t], len = tasks[2 * t + 1];
                int64_t e = pos[s];
                const int64_t eend = e + batch < len ? e + batch : len;
                for (; e < eend; ++e) {
                    const uint64_t base = (uint64_t)(uint32_t)col[kbeg + e] * (uint64_t)row_lines + (uint64_t)p * lines_per_pass;
                    for (int l = 0; l < lines_per_pass; ++l) {
                        if (cache_access(&c, base + l)) ++hits; else ++misses;
                    }
                }
                pos[s] = e;
                if (e >= len) {
                    if (next < ntasks) { cur[s] = next++; pos[s] = 0; }
                    else { cur[s] = -1; --active; }
                }
            }
        }
    }
    *hits_out = hits; *misses_out = misses;
    free(c.tag); free(c.stamp); free(cur); free(pos);
    return 0;
}
