/* The Go shim's twin in plain C: go/bigseqkit/bsk_cgo.go cannot be compiled here (no Go toolchain), and
 * it is -- next to the command line -- the one consumer of bsk_comm_init_all + one thread per rank.  This program drives the
 * EXACT call sequence of StatsN / GrepCountN / RmDupN from N pthreads, through include/bsk.h alone (no HIP, no C++):
 *     bsk_comm_init_all(devices)                       NewComms
 *     per rank, on its own thread:                     onEveryDevice
 *         bsk_create, the rank's own work              (everything that can fail on this rank alone)
 *         bsk_count_allreduce(failed ? 1 : 0)          Comms.agree: EVERY rank enters it, also one that failed
 *         the collective that carries data             bsk_stats_collect_reduced / bsk_count_allreduce / bsk_rmdup_dist_run
 * usage: ranks <file.fq> <devices, e.g. 0,0,0> <outdir> [bad-rank]
 *   writes <outdir>/stats.txt (rank 0's table), <outdir>/grepc.txt, <outdir>/rmdup.<rank>; with `bad-rank` that rank is given a
 *   shard that is no FASTQ: every rank must come back with an error and nobody may hang.
 * References: /root/reference/bigseqkit/stats.go:91 (Reduce), grep.go:175, rmdup.go:97 (GroupByKey). */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bsk.h"

#define MAXR 16

typedef struct {
    int rank, world, device, bad;
    bsk_comm* comm;
    const uint8_t* shard;
    size_t n;
    const char* outdir;
    char err[512];
    int failed;
} Rank;

static int agree(Rank* r, int own_failed) { /* Comms.agree of the Go shim */
    uint64_t bad = own_failed ? 1 : 0;
    if (bsk_count_allreduce(r->comm, &bad, NULL) != BSK_OK) { snprintf(r->err, sizeof r->err, "agree: %s", bsk_comm_error(r->comm)); return -1; }
    return (int)bad;
}

static void fail(Rank* r, const char* what, const char* why) {
    if (!r->failed) snprintf(r->err, sizeof r->err, "%s: %s", what, why ? why : "?");
    r->failed = 1;
}

static void* work(void* arg) {
    Rank* r = (Rank*)arg;
    char path[1024];
    /* ---- StatsN ---------------------------------------------------------------------------------------------------- */
    {
        bsk_ctx* ctx = NULL;
        int own = 0;
        if (bsk_create("Stats", "{\"All\":true,\"Tabular\":true}", r->device, &ctx) != BSK_OK) { fail(r, "bsk_create", bsk_global_error()); own = 1; }
        if (!own && r->n && bsk_stats_run(ctx, r->shard, r->n, 0, BSK_FORMAT_FASTQ, r->rank, NULL, NULL) != BSK_OK) { fail(r, "bsk_stats_run", bsk_last_error(ctx)); own = 1; }
        const int bad = agree(r, own);
        if (bad == 0) {
            static const size_t CAP = 1 << 16;
            int64_t* keys = (int64_t*)malloc(CAP * 8);
            int64_t* vals = (int64_t*)malloc(CAP * 8);
            size_t cnt = 0;
            if (bsk_stats_collect_reduced(ctx, r->comm, NULL, NULL, keys, vals, CAP, &cnt) != BSK_OK) fail(r, "bsk_stats_collect_reduced", bsk_last_error(ctx));
            else if (r->rank == 0) {
                bsk_statinfo info;
                char table[8192];
                if (bsk_stats_finalize(ctx, keys, vals, cnt, &info) != BSK_OK || bsk_stats_string(ctx, "input0", "N/A", &info, table, sizeof table) != BSK_OK)
                    fail(r, "bsk_stats_string", bsk_last_error(ctx));
                else {
                    snprintf(path, sizeof path, "%s/stats.txt", r->outdir);
                    FILE* f = fopen(path, "wb");
                    if (f) { fputs(table, f); fclose(f); }
                }
            }
            free(keys);
            free(vals);
        } else if (bad > 0 && !own) fail(r, "stats", "another rank failed before the collective");
        if (ctx) bsk_destroy(ctx);
    }
    /* ---- GrepCountN ------------------------------------------------------------------------------------------------ */
    {
        bsk_ctx* ctx = NULL;
        int own = 0;
        uint64_t cnt = 0;
        if (bsk_create("Grep", "{\"BySeq\":true,\"Pattern\":[\"ACGTTGCA\"],\"Count\":true}", r->device, &ctx) != BSK_OK) { fail(r, "bsk_create", bsk_global_error()); own = 1; }
        if (!own) {
            bsk_out o;
            if (bsk_grep_run(ctx, r->shard, r->n, 0, BSK_FORMAT_FASTQ, r->rank, NULL, &o) != BSK_OK || bsk_grep_last_count(ctx, &cnt) != BSK_OK) { fail(r, "bsk_grep_run", bsk_last_error(ctx)); own = 1; cnt = 0; }
        }
        const int bad = agree(r, own);
        if (bad == 0) {
            if (bsk_count_allreduce(r->comm, &cnt, NULL) != BSK_OK) fail(r, "bsk_count_allreduce", bsk_comm_error(r->comm));
            else if (r->rank == 0) {
                snprintf(path, sizeof path, "%s/grepc.txt", r->outdir);
                FILE* f = fopen(path, "wb");
                if (f) { fprintf(f, "%llu", (unsigned long long)cnt); fclose(f); }
            }
        } else if (bad > 0 && !own) fail(r, "grep -C", "another rank failed before the collective");
        if (ctx) bsk_destroy(ctx);
    }
    /* ---- RmDupN ---------------------------------------------------------------------------------------------------- */
    {
        bsk_ctx* ctx = NULL;
        void* dev = NULL;
        int own = 0;
        if (bsk_create("RmDup", "{\"BySeq\":true}", r->device, &ctx) != BSK_OK) { fail(r, "bsk_create", bsk_global_error()); own = 1; }
        if (!own) {
            bsk_device_select(r->device);
            dev = bsk_device_alloc(r->n + 1);
            if (!dev) { fail(r, "bsk_device_alloc", bsk_global_error()); own = 1; }
            else if (r->n && bsk_device_copy(dev, r->shard, r->n, BSK_COPY_H2D) != BSK_OK) { fail(r, "bsk_device_copy", bsk_global_error()); own = 1; }
        }
        const int bad = agree(r, own);
        if (bad == 0) {
            bsk_out out;
            /* (from here on bsk_rmdup_dist_run carries every phase's outcome with its next collective itself) */
            if (bsk_rmdup_dist_run(ctx, r->comm, dev, r->n, BSK_FORMAT_FASTQ, NULL, &out) != BSK_OK) fail(r, "bsk_rmdup_dist_run", bsk_last_error(ctx));
            else {
                uint8_t* host = (uint8_t*)malloc(out.len ? out.len : 1);
                if (out.len && bsk_out_to_host(ctx, &out, host, out.len) != BSK_OK) fail(r, "bsk_out_to_host", bsk_last_error(ctx));
                else {
                    snprintf(path, sizeof path, "%s/rmdup.%d", r->outdir, r->rank);
                    FILE* f = fopen(path, "wb");
                    if (f) { fwrite(host, 1, out.len, f); fclose(f); }
                    uint64_t a = 0, b = 0, fl = 0;
                    bsk_rmdup_dist_stats(ctx, &a, &b, &fl);
                    snprintf(path, sizeof path, "%s/pairs.%d", r->outdir, r->rank);
                    f = fopen(path, "wb");
                    if (f) { fprintf(f, "%llu %llu %llu %llu", (unsigned long long)a, (unsigned long long)b, (unsigned long long)fl, (unsigned long long)out.records); fclose(f); }
                }
                free(host);
            }
        } else if (bad > 0 && !own) fail(r, "rmdup", "another rank failed before the collective");
        if (dev) bsk_device_free(dev);
        if (ctx) bsk_destroy(ctx);
    }
    return NULL;
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s file.fq devices outdir [bad-rank]\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    fseek(f, 0, SEEK_END);
    const size_t n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t* data = (uint8_t*)malloc(n + 1);
    if (fread(data, 1, n, f) != n) { fprintf(stderr, "short read\n"); return 2; }
    fclose(f);
    int devices[MAXR], world = 0;
    for (char* tok = strtok(argv[2], ","); tok && world < MAXR; tok = strtok(NULL, ",")) devices[world++] = atoi(tok);
    const int bad_rank = argc > 4 ? atoi(argv[4]) : -1;
    /* record-aligned shards: the ReadFixer rule of the library itself */
    size_t cuts[MAXR + 1];
    cuts[0] = 0;
    for (int k = 1; k < world; ++k) {
        size_t at = 0;
        if (bsk_find_record_start(data, n, n * (size_t)k / (size_t)world, BSK_FORMAT_FASTQ, &at) != BSK_OK) return 2;
        cuts[k] = at < cuts[k - 1] ? cuts[k - 1] : at;
    }
    cuts[world] = n;
    bsk_comm* comms[MAXR];
    if (bsk_comm_init_all(world, devices, comms) != BSK_OK) { fprintf(stderr, "bsk_comm_init_all: %s\n", bsk_comm_error(NULL)); return 1; }
    Rank R[MAXR];
    pthread_t th[MAXR];
    static const uint8_t garbage[] = "@x\nAC\n+\nIII\n@y\nACGT\n+\nII\n";
    for (int k = 0; k < world; ++k) {
        memset(&R[k], 0, sizeof R[k]);
        R[k].rank = k; R[k].world = world; R[k].device = devices[k]; R[k].comm = comms[k]; R[k].outdir = argv[3];
        R[k].shard = data + cuts[k]; R[k].n = cuts[k + 1] - cuts[k];
        if (k == bad_rank) { R[k].shard = garbage; R[k].n = sizeof garbage - 1; }
        pthread_create(&th[k], NULL, work, &R[k]);
    }
    int failed = 0;
    for (int k = 0; k < world; ++k) {
        pthread_join(th[k], NULL);
        if (R[k].failed) { fprintf(stderr, "rank %d: %s\n", k, R[k].err); ++failed; }
    }
    for (int k = 0; k < world; ++k) bsk_comm_destroy(comms[k]);
    free(data);
    printf("ranks=%d failed=%d\n", world, failed);
    return failed ? 1 : 0;
}
