/*
 * mfsgd.h -- C-ABI of libmfsgd.so, the MI355X (gfx950) matrix-factorisation
 * SGD trainer.  This is the drop-in boundary: what a JNI / ctypes / cgo stub
 * binds.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Reference interface replaced: NONE EXISTS.  /root/reference/README.md:1-2 is
 * the whole reference repository (a title and a course attribution); it has no
 * class, no FFI and no operator interface.  The surface below follows
 * SURVEY.md section 8b, which derives it from BASELINE.json's north_star
 * ("keeping the Java MatrixFactorizationSGD train()/predict() surface ...
 * through a thin JNI C-ABI").  Each entry point names the Java method it backs
 * (matrixfactorizationsgd.java_amd/java/MatrixFactorizationSGD.java).
 *
 * Conventions
 *  - every function returns MFSGD_OK (0) or a negative mfsgd_status;
 *    the message is available from mfsgd_last_error();
 *  - no C++ exception and no HIP error crosses this boundary;
 *  - the caller owns every host array passed in or out; the library copies;
 *  - the handle owns all device memory; mfsgd_destroy() frees it;
 *  - a handle is not thread-safe; distinct handles are independent;
 *  - there is NO CPU fallback: compute entry points return
 *    MFSGD_ERR_NO_DEVICE when no gfx950 device is usable.  Host-only entry
 *    points (create, set_ratings = schedule construction, schedule queries)
 *    work without a GPU.
 */
#ifndef MFSGD_H
#define MFSGD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFSGD_ABI_VERSION 3 /* 3 (round 3): + mfsgd_part_settle, mfsgd_dsgd_plan_ex, mfsgd_dsgd_stats; solo-record word order in the debug arrays */

typedef enum mfsgd_status {
    MFSGD_OK = 0,
    MFSGD_ERR_INVALID_ARG = -1,
    MFSGD_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture          */
    MFSGD_ERR_HIP = -3,         /* a HIP runtime call failed (see last_error)  */
    MFSGD_ERR_OOM = -4,         /* host or device allocation failed            */
    MFSGD_ERR_STATE = -5,       /* call order violated (e.g. train before set) */
    MFSGD_ERR_UNSUPPORTED = -6, /* e.g. k > MFSGD_MAX_K                        */
    MFSGD_ERR_SCHEDULE = -7     /* schedule could not be built (bad index, 32-bit overflow) */
} mfsgd_status;

#define MFSGD_MAX_K 256

typedef struct mfsgd_handle mfsgd_handle;

/* Hyper-parameters and geometry.  Zero in an "auto" field selects the default.
 * Java: constructor MatrixFactorizationSGD(users, items, k, lr, lambda, seed). */
typedef struct mfsgd_config {
    int32_t n_users;      /* U: rows of P held by this handle                       */
    int32_t n_items;      /* I: rows of Q (global item count)                       */
    int32_t k;            /* latent dimension, 1..MFSGD_MAX_K                       */
    float lr;             /* learning rate                                          */
    float lambda;         /* L2 regularisation, shared by P and Q                   */
    int32_t device;       /* HIP device ordinal                                     */
    int32_t blocks;       /* B: user blocks = item tiles per side; 0 = auto         */
    int32_t waves;        /* W: waves per workgroup (1,2,4,8); 0 = auto             */
    int32_t n_parts;      /* DSGD: item partitions (= GPUs); 0 or 1 = single device */
    int32_t host_threads; /* schedule-construction threads; 0 = hardware threads    */
    int32_t flags;        /* MFSGD_FLAG_*                                           */
    int32_t reserved[5];  /* must be zero                                           */
} mfsgd_config;

#define MFSGD_FLAG_NO_GRAPH 1     /* launch eagerly instead of replaying a hipGraph            */
#define MFSGD_FLAG_HOST_INGEST 4   /* bucket the ratings on the host even when a GPU is present     */
#define MFSGD_FLAG_DEVICE_INGEST 8 /* ... on the GPU even for small rating sets (default: >= 2^20)  */
#define MFSGD_FLAG_HOST_PACK 32     /* pack the cells on the host even when the device could (tests: both build the same bytes) */
#define MFSGD_FLAG_NO_SOLO 16      /* schedule without solo runs (A/B measurements, tests)           */
#define MFSGD_FLAG_ROUND_LAUNCH 2 /* one kernel launch per round instead of the persistent      */
                                  /* epoch kernel (which hands item tiles between workgroups)   */

/* What the scheduler built; for tests, bench.py's roofline arithmetic, DESIGN. */
typedef struct mfsgd_schedule_info {
    int64_t nnz;          /* ratings in this part                                   */
    int32_t part;         /* item partition this describes                          */
    int32_t blocks;       /* B                                                      */
    int32_t waves;        /* W                                                      */
    int32_t group_lanes;  /* L: lanes per rating (row bytes / 16)                   */
    int32_t slots;        /* G = 64 / L ratings per wave step                       */
    int32_t kp;           /* padded row length in floats (device stride)            */
    int32_t rounds;       /* rounds per epoch (= B)                                 */
    int32_t lds_bytes;    /* dynamic LDS requested per workgroup                    */
    int64_t total_steps;  /* wave steps over all cells                              */
    int64_t total_rows;   /* factor rows gathered (and scattered) per epoch         */
    int64_t max_cell_nnz;
    int64_t max_cell_rows;  /* rows of the largest chunk */
    int64_t max_cell_steps; /* critical path of the slowest cell (sum over sub-rounds of max wave steps) */
    int64_t sum_round_steps; /* sum over rounds of the slowest cell's critical path */
    double build_seconds;
    int32_t swapped;      /* 1: roles exchanged (users on the kernel's forwarding side): in the   */
    int32_t device_ingest; /* 1: degree histograms and bucket order were computed on the GPU; 2: the cells */
                           /* were packed there too (rows / entries / order never existed on the host)  */
    int64_t chunks;       /* chunk descriptors (>= blocks*blocks: one per cell + extra chunks)    */
    int64_t split_cells;  /* cells cut into more than one chunk because they exceed the LDS       */
} mfsgd_schedule_info;

/* ---- lifetime ------------------------------------------------------------- */
int mfsgd_abi_version(void);
/* Number of usable gfx950 devices; 0 (and MFSGD_OK) when there are none. */
int mfsgd_device_count(int32_t* out);
/* Java: constructor.  Does not touch the GPU. */
int mfsgd_create(const mfsgd_config* cfg, mfsgd_handle** out);
/* Java: close(). NULL is allowed. */
void mfsgd_destroy(mfsgd_handle* h);
/* Message of the last failure on this handle (h == NULL: of the last failed
 * mfsgd_create on this thread).  Never NULL; valid until the next call. */
const char* mfsgd_last_error(const mfsgd_handle* h);

/* ---- ratings -> schedule (host) -------------------------------------------
 * Java: first half of train(int[] u, int[] i, float[] r, int epochs).
 * COO triples; 0 <= u < n_users, 0 <= i < n_items.  Buckets the ratings into
 * B x B (user block, item tile) cells per item partition and packs every cell
 * into conflict-free wave steps.  Works without a GPU; with one (and >= 2^20
 * ratings) the streaming passes -- degree histograms, bucket order -- run on it,
 * with identical results.  Replaces any earlier ratings -- unless they ARE the
 * earlier ratings: the same triples again (length and a 128-bit hash of every
 * byte of the three arrays) keep the schedules and their device copies, so a host
 * may hand train() the same arrays on every call at the cost of one pass over them. */
int mfsgd_set_ratings(mfsgd_handle* h, const int32_t* u, const int32_t* i, const float* r,
                      int64_t nnz);

/* ---- factors --------------------------------------------------------------
 * Host layout is dense row-major: P is n_users x k, Q is n_items x k.
 * init: java.util.Random(seed).nextFloat() * (float)(1/sqrt(k)), P then Q.
 * With n_parts > 1 only P (and nothing of Q) lives in the handle: Q travels
 * in caller-owned device blocks (mfsgd_part_* below); pass Q = NULL here.    */
int mfsgd_init_factors(mfsgd_handle* h, int64_t seed);
int mfsgd_set_factors(mfsgd_handle* h, const float* P, const float* Q);
int mfsgd_get_factors(mfsgd_handle* h, float* P, float* Q);

/* ---- the hot path ---------------------------------------------------------
 * Java: second half of train(...).  Runs `epochs` passes over the schedule on
 * the device.  If rmse_per_epoch != NULL it receives the RMSE after each epoch
 * (one extra read-only pass per epoch).  Blocks until the device is idle.    */
int mfsgd_train(mfsgd_handle* h, int32_t epochs, double* rmse_per_epoch);
/* RMSE over the stored ratings with the current factors. */
int mfsgd_rmse(mfsgd_handle* h, double* out);
/* Java: predict(int u, int i) / predict(int[] u, int[] i). */
int mfsgd_predict(mfsgd_handle* h, const int32_t* u, const int32_t* i, float* out, int64_t n);

/* Top-N scoring: for each of n_users users the topn items with the largest
 * dot(P[u], Q[i]) -- the same bits mfsgd_predict() returns -- best first, ties by the
 * smaller item index.  out_items / out_scores: n_users x topn, row-major.          */
int mfsgd_recommend(mfsgd_handle* h, const int32_t* users, int32_t n_users, int32_t topn, int32_t* out_items,
                    float* out_scores);

/* Timed variant used by bench.py: runs `epochs` training passes bracketed by
 * HIP events on the handle's stream and returns the elapsed device time and
 * the number of sgd-round kernel launches inside it.  No RMSE pass.          */
int mfsgd_train_timed(mfsgd_handle* h, int32_t epochs, double* elapsed_ms, int64_t* launches);

/* ---- data formats either side of the path (host; csrc/io.cpp) ----------------
 * Rating files of the datasets the configs are shaped after.  Ids in the file are
 * arbitrary: they are compacted to dense indices (rank among the distinct ids,
 * ascending); the original ids are returned as tables.  Errors of these functions
 * are reported by mfsgd_io_last_error() (they have no handle).                   */
#define MFSGD_FMT_AUTO 0
#define MFSGD_FMT_ML_TSV 1  /* MovieLens-100K u.data: user \t item \t rating \t timestamp   */
#define MFSGD_FMT_ML_DAT 2  /* MovieLens-1M/10M ratings.dat: user::movie::rating::time   */
#define MFSGD_FMT_ML_CSV 3  /* MovieLens-20M/25M ratings.csv: userId,movieId,rating,time */
#define MFSGD_FMT_NETFLIX 4 /* Netflix Prize: "movieId:" then "customerId,rating,date"   */
typedef struct mfsgd_ratings_file mfsgd_ratings_file;
int mfsgd_ratings_file_open(const char* path, int32_t format, mfsgd_ratings_file** out);
int mfsgd_ratings_file_info(const mfsgd_ratings_file* f, int64_t* nnz, int32_t* n_users, int32_t* n_items);
/* Any pointer may be NULL.  u, i, r: nnz entries; user_ids / item_ids: n_users / n_items. */
int mfsgd_ratings_file_read(const mfsgd_ratings_file* f, int32_t* u, int32_t* i, float* r,
                            int64_t* user_ids, int64_t* item_ids);
void mfsgd_ratings_file_close(mfsgd_ratings_file* f);
const char* mfsgd_io_last_error(void);
/* Factor files: "MFSGDF01", int32 U, I, k, 0, then P (U x k) and Q (I x k), fp32 little endian. */
int mfsgd_get_dims(const mfsgd_handle* h, int32_t* n_users, int32_t* n_items, int32_t* k);
int mfsgd_save_factors(mfsgd_handle* h, const char* path);
int mfsgd_load_factors(mfsgd_handle* h, const char* path);

/* ---- schedule introspection (host) ---------------------------------------- */
int mfsgd_get_schedule_info(const mfsgd_handle* h, int32_t part, mfsgd_schedule_info* out);
/* Canonical sequential order of partition `part`: order[j] is the index (into
 * the arrays given to set_ratings) of the j-th rating; length = info.nnz.
 * cell_ptr has rounds*blocks+1 entries: cell b of round rd covers
 * order[cell_ptr[rd*blocks+b] .. cell_ptr[rd*blocks+b+1]).                    */
int mfsgd_get_order(const mfsgd_handle* h, int32_t part, int64_t* order, int64_t* cell_ptr);

/* Diagnostic (not part of the Java surface): the device-facing schedule arrays of a
 * partition, so that tests can replay the kernel's exact LDS access order on the CPU.
 * cells: n_cells x 8 words {row_off, ent_off, n_steps, nu | ni << 16, next chunk, flags, 2 reserved} (flags bit 0:
 * the cell's tile is ONE item row in every cell -- it is handed on through the tile's mailbox): chunk
 * descriptors -- the first blocks*blocks are the first chunk of each cell, a cell too large for
 * the LDS continues through `next` (0 = last chunk) into the descriptors behind them;
 * subs: n_subs x 2 words {off | solo steps << 16, general steps | run steps << 16};
 * entries: n_entries x 4 words {p addr | q addr << 16 | flag << 31 (16-byte units), rating bits,
 * bits of lr*rating, bits of the slot's decay factor}.  A sub-cell's solo run follows its run steps
 * and two idle steps (padded to whole steps): header {slots_0, 0, 0, 0}, then per step {slots of the NEXT step,
 * 0xFFFFFFFF (mailbox), bits of lr*rating, rating bits}, then a terminator; the slots behind the last step
 * address an all-zero row.                                                                      */
int mfsgd_debug_schedule_sizes(const mfsgd_handle* h, int32_t part, int64_t* n_cells, int64_t* n_rows,
                               int64_t* n_subs, int64_t* n_entries);
int mfsgd_debug_get_schedule(const mfsgd_handle* h, int32_t part, uint32_t* cells, uint32_t* rows,
                             uint32_t* subs, uint32_t* entries);

/* Diagnostic (not part of the Java surface): runs ONE epoch with the persistent kernel and
 * returns, per workgroup, 16 words: shader cycles wave 0 spent in (0) draining its previous
 * stores and issuing prefetch + own-row gather, (1) waiting for the tile, (2) the barrier after
 * it, (3) the tile gather, (4) the ratings, (5) storing + publishing the tile, (6) storing its
 * own rows; (7) the longest single ratings phase (its slowest cell); (8..14) the seven phases of
 * the pass that contained it; (15) unused.  out must hold blocks x 16 words.  It DOES apply the epoch. */
int mfsgd_debug_epoch_profile(mfsgd_handle* h, uint64_t* out, int32_t* n_workgroups);

/* Diagnostic: out4 = {times a persistent launch found its workgroups not co-resident and the library
 * switched that partition to round launches, partitions currently on the persistent kernel, cached
 * training graphs, schedule builds: mfsgd_set_ratings calls that did not find the same triples
 * already in place}.                                                                              */
int mfsgd_debug_counters(const mfsgd_handle* h, int64_t* out4);

/* Diagnostic (not part of the Java surface): occupies the LDS of all but four CUs for `milliseconds` (<= 5000)
 * with a spinning kernel on a side stream, asynchronously -- what a foreign kernel sharing the GPU
 * looks like to the persistent epoch kernel.  Tests use it to force the "workgroups not co-resident"
 * path: the epoch kernel gives up before touching anything and the epoch runs as round launches. */
int mfsgd_debug_occupy(mfsgd_handle* h, int32_t milliseconds);

/* Diagnostic (not part of the Java surface): runs training round `round` once with
 * phase stamps; out receives blocks x 6 values per workgroup: shader-clock at
 * start, after gather, after the rating steps, after scatter, then the 100 MHz
 * constant clock at start and at end; after those blocks x 6 words follow
 * blocks x W x W x 4 words: per (wave, sub-round) cycles in the general loop,
 * cycles in the run loop, general steps, run steps.  It DOES apply
 * that round's updates.  Single-partition handles only.                        */
int mfsgd_debug_round_stamps(mfsgd_handle* h, int32_t part, int32_t round, uint64_t* out);

/* ---- DSGD building blocks (n_parts > 1) ------------------------------------
 * The global partitioner (host, no GPU): ONE rating set over n_users x n_items is cut for
 * n_parts devices -- users into n_parts contiguous ranges balanced by rating count (device g
 * holds the P rows of users [user_begin[g], user_begin[g+1]) and those users' ratings), items
 * into n_parts partitions balanced by rating count (longest-processing-time-first; items nobody
 * rated are dealt out to even the row counts).  A pure function of the two degree arrays, so
 * every rank computes the same plan from the same degrees (a rank that only sees its own ratings
 * all-reduces the item histogram first).  user_begin: n_parts + 1 entries; item_part: n_items.  */
int mfsgd_dsgd_plan(const int64_t* deg_user, const int64_t* deg_item, int32_t n_users, int32_t n_items,
                    int32_t n_parts, int32_t* user_begin, int32_t* item_part);
/* The same for a job of `world` ranks holding `parts_per_rank` item partitions at a time, at rank k: user_begin has
 * world + 1 entries, item_part n_items entries in [0, world * parts_per_rank).
 * chain_crit = 0: items dealt longest-processing-time-first by rating count -- the heaviest items end up one per
 * partition, the partitions take the same time, and the ring's epoch (world x the slowest partition: a Q block is
 * trained by one rank after the other) is as short as the heaviest item's own chain allows.  This is what
 * mfsgd_dsgd_plan does and what a ring of GPUs wants.
 * chain_crit > 0 (e.g. 0.3): CHAIN-AWARE -- the chain-critical items (those whose chain of dependent updates on one
 * rank reaches chain_crit of a work-bound sub-epoch) are packed, heaviest together, into as few partitions as the
 * rating-count balance allows, the rest dealt LPT over the others.  That minimises the SUM over the partitions of
 * their heaviest items' chains -- what ONE device pays when it runs the partitions back to back (virtual devices) --
 * at the price of unequal partition times (measured: DESIGN.md section 6); not for a ring.
 * info4 (nullable): {sum over the partitions of their heaviest item's rating count, chain-critical items,
 * partitions filled sequentially, the threshold}.   Java: MatrixFactorizationSGD.plan(degUser, degItem, nParts). */
int mfsgd_dsgd_plan_ex(const int64_t* deg_user, const int64_t* deg_item, int32_t n_users, int32_t n_items,
                       int32_t world, int32_t parts_per_rank, int32_t k, float chain_crit, int32_t* user_begin,
                       int32_t* item_part, int64_t* info4);
/* Installs an item -> partition map (n_items entries in [0, n_parts)) on a handle with
 * n_parts > 1, before mfsgd_set_ratings; item i is then row (number of smaller item ids in the
 * same partition) of that partition's Q block.  NULL restores the default below.            */
int mfsgd_set_item_partition(mfsgd_handle* h, const int32_t* item_part);
/* The map in force: partition and block row of every item (either pointer may be NULL). */
int mfsgd_get_item_partition(const mfsgd_handle* h, int32_t* item_part, int32_t* item_row);
/* Default map: item i belongs to partition i % n_parts and is row i / n_parts of that
 * partition's Q block.  A Q block is a caller-owned DEVICE buffer of
 * mfsgd_part_rows() x kp floats (kp from schedule_info), so that the host side
 * can move it between GPUs (RCCL send/recv) without this library knowing.
 * `stream` is the hipStream_t the caller orders its use of the block on; it is
 * used as given (NULL = HIP's null stream, which is what torch's default stream
 * is), never replaced by a private stream.                                     */
int mfsgd_part_rows(const mfsgd_handle* h, int32_t part, int32_t* rows);
/* Fill a HOST buffer (rows x kp, zero padded) with the initial values the
 * single-device init would give these items for this seed and this U:
 * the stream position of Q row i is (U_total + i) * k.                        */
int mfsgd_part_init_q(const mfsgd_handle* h, int32_t part, int64_t seed, int64_t u_total,
                      float* q_block_host);
/* One DSGD sub-epoch: every rating of this handle's users whose item is in
 * `part`, against q_block_dev.  Asynchronous on `stream`.  Calls for different
 * partitions of ONE handle must not overlap in time: they update the same P rows. */
int mfsgd_part_train(mfsgd_handle* h, int32_t part, float* q_block_dev, void* stream);
/* Sum of squared errors of the same ratings (fp64), synchronous. */
int mfsgd_part_sse(mfsgd_handle* h, int32_t part, const float* q_block_dev, void* stream,
                   double* sse);
/* Seeds P when this handle holds users [u_offset, u_offset + n_users) of a
 * larger problem: stream position of P row u is (u_offset + u) * k.           */
int mfsgd_init_p_offset(mfsgd_handle* h, int64_t seed, int64_t u_offset);
/* The recovery point of an asynchronous sub-epoch: waits for `stream`; if the LAST mfsgd_part_train of this
 * partition found its persistent launch not co-resident (another kernel held CUs; the launch then changed
 * nothing), the sub-epoch is trained now, as one launch per round on the same stream, and waited for;
 * *rerun (nullable) = 1 then and the partition stays on round launches.  Call it with the same block
 * BEFORE the block is passed on; csrc/dsgd.cpp does so where an exchange may share the GPU with training. */
int mfsgd_part_settle(mfsgd_handle* h, int32_t part, float* q_block_dev, void* stream, int32_t* rerun);
/* Waits for `stream` and reports whether a training launch of this partition gave up on a
 * hand-off since the last check (MFSGD_ERR_HIP then: the factors are invalid).  mfsgd_part_train
 * is asynchronous and cannot report that itself.                                             */
int mfsgd_part_sync(mfsgd_handle* h, int32_t part, void* stream);
/* Partition count, padded row length (floats) and device ordinal of a handle. */
int mfsgd_get_parts(const mfsgd_handle* h, int32_t* n_parts, int32_t* kp, int32_t* device);

/* ---- DSGD driver: the ring under the C-ABI (csrc/dsgd.cpp) -----------------------------------
 * One process per GPU.  Each rank creates a handle with n_parts = world * m (m >= 1 partitions held
 * at a time), installs the item map (mfsgd_set_item_partition) if it uses a plan, gives it ITS
 * users' ratings and seeds P (mfsgd_init_p_offset); then
 *     rank 0: mfsgd_dsgd_unique_id(id);  every rank receives the same 128 bytes (any transport)
 *     mfsgd_dsgd_create(h, rank, world, id, &d)   -- collective (ncclCommInitRank)
 *     mfsgd_dsgd_init_q(d, seed, u_total)         -- seeds the Q blocks this rank holds first
 *     mfsgd_dsgd_train(d, epochs, rmse)           -- collective
 * An epoch is `world` sub-epochs: train the m partitions of the group held (one after another:
 * they share this rank's users), pass each block to rank - 1 as soon as ITS training has finished
 * and receive the next one from rank + 1: ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on
 * a communication stream, ordered against the compute stream with events, no host
 * synchronisation inside an epoch.  RCCL is bound at run time (librccl.so.1, or
 * MFSGD_RCCL_LIBRARY); without it these calls return MFSGD_ERR_UNSUPPORTED.
 * Rehearsal on one GPU: if MFSGD_DSGD_TRANSPORT=shm is set when mfsgd_dsgd_unique_id runs, the id
 * names a POSIX shared-memory segment and the ranks (processes of one host) move the blocks
 * through it instead of RCCL -- same results, host-copy speed, for testing a host's multi-process
 * logic where RCCL cannot run (two ranks on one GPU).
 * Java: MatrixFactorizationSGD.trainDistributed(...).               */
typedef struct mfsgd_dsgd mfsgd_dsgd;
#define MFSGD_DSGD_ID_BYTES 128
int mfsgd_dsgd_unique_id(void* id_out /* MFSGD_DSGD_ID_BYTES */);
int mfsgd_dsgd_create(mfsgd_handle* h, int32_t rank, int32_t world, const void* id, mfsgd_dsgd** out);
void mfsgd_dsgd_destroy(mfsgd_dsgd* d);
/* d == NULL: message of the last failed mfsgd_dsgd_create / mfsgd_dsgd_unique_id on this thread. */
const char* mfsgd_dsgd_last_error(const mfsgd_dsgd* d);
/* Seeds the blocks of the group this rank holds first (partitions rank*m .. rank*m + m - 1) as
 * the single-device init would: stream position of Q row i is (u_total + i) * k.             */
int mfsgd_dsgd_init_q(mfsgd_dsgd* d, int64_t seed, int64_t u_total);
/* Slot j (0 <= j < m) of the group currently held -- between epochs that is the home group:
 * dense rows x k host arrays, rows = mfsgd_part_rows of that partition.                      */
int mfsgd_dsgd_set_q(mfsgd_dsgd* d, int32_t j, const float* block_host);
int mfsgd_dsgd_get_q(mfsgd_dsgd* d, int32_t j, int32_t* part, int32_t* rows, float* block_host);
/* `epochs` DSGD epochs; rmse_per_epoch (nullable) receives the global RMSE after each (one
 * read-only rotation and a 2-double all-reduce).  Blocks until the device is idle.           */
int mfsgd_dsgd_train(mfsgd_dsgd* d, int32_t epochs, double* rmse_per_epoch);
int mfsgd_dsgd_rmse(mfsgd_dsgd* d, double* out);
/* bench.py: `epochs` epochs bracketed by HIP events on the compute stream (the last blocks'
 * arrival included); no RMSE pass.                                                            */
int mfsgd_dsgd_train_timed(mfsgd_dsgd* d, int32_t epochs, double* elapsed_ms);
/* All-reduce of two doubles over the ranks (op 0 = sum, 1 = max): what a host needs for global
 * counts and max-over-ranks timings without a second communication library.                  */
int mfsgd_dsgd_allreduce(mfsgd_dsgd* d, double* values2, int32_t op);
/* Counters of this rank's ring since mfsgd_dsgd_create: out4 = {sub-epoch trainings enqueued, of those run with
 * the recovery point (mfsgd_part_settle before the block leaves), of those re-run as round launches because the
 * persistent launch was not co-resident, bytes sent}.                                                         */
int mfsgd_dsgd_stats(const mfsgd_dsgd* d, int64_t* out4);

#ifdef __cplusplus
}
#endif
#endif
