/*
 * MatrixFactorizationSGD.java -- the Java host surface over libmfsgd.so.
 *
 * UNCOMPILED AND UNTESTED: neither this container nor the GPU box has a JDK
 * (javac/java absent, no jni.h), see DESIGN.md section 8.  The reference
 * repository contains no Java source either (/root/reference/README.md:1-2 is
 * all of it); the surface is the one BASELINE.json names -- a class
 * MatrixFactorizationSGD with train()/predict() -- with the parameter lists
 * SURVEY.md section 8b proposes.  Every native method is one call into the
 * C-ABI of include/mfsgd.h through jni/mfsgd_jni.cpp.
 */
public final class MatrixFactorizationSGD implements AutoCloseable {
    static {
        System.loadLibrary("mfsgd_jni"); // libmfsgd_jni.so, linked against libmfsgd.so
    }

    private long handle; // mfsgd_handle*
    private long ring;   // mfsgd_dsgd* (trainDistributed), 0 until the first call
    private int slots;   // item partitions this rank holds at a time (nParts / world)
    private final int users, items, k, nParts;
    private final long seed;
    private boolean initialised;

    public MatrixFactorizationSGD(int users, int items, int k, float lr, float lambda, long seed) {
        this.users = users;
        this.items = items;
        this.k = k;
        this.seed = seed;
        this.nParts = 1;
        this.handle = nativeCreate(users, items, k, lr, lambda, /*device*/ 0, /*nParts*/ 0);
    }

    /**
     * One rank of a DSGD job (one JVM process per GPU): this rank's {@code users} P rows, the GLOBAL item count, and
     * {@code nParts = world * partsPerRank} item partitions whose Q blocks travel round the ring of ranks.
     */
    public MatrixFactorizationSGD(int users, int items, int k, float lr, float lambda, long seed, int device, int nParts) {
        if (nParts < 2) throw new IllegalArgumentException("a distributed handle has at least two item partitions");
        this.users = users;
        this.items = items;
        this.k = k;
        this.seed = seed;
        this.nParts = nParts;
        this.handle = nativeCreate(users, items, k, lr, lambda, device, nParts);
    }

    /** Runs {@code epochs} SGD passes over the ratings; returns the RMSE after each epoch. */
    public double[] train(int[] u, int[] i, float[] r, int epochs) {
        if (u.length != i.length || u.length != r.length) throw new IllegalArgumentException("length mismatch");
        // The library recognises the same triples again (length + a 128-bit hash of every byte) and keeps
        // its schedules, so repeated train() calls on one rating set pay for one pass over the arrays, not
        // for a new schedule.  (Comparing array identity here would miss in-place edits of the arrays.)
        nativeSetRatings(handle, u, i, r);
        if (!initialised) {
            nativeInitFactors(handle, seed); // java.util.Random(seed).nextFloat()/sqrt(k), P then Q
            initialised = true;
        }
        double[] rmse = new double[epochs];
        nativeTrain(handle, epochs, rmse);
        return rmse;
    }

    public float predict(int u, int i) {
        return predict(new int[] {u}, new int[] {i})[0];
    }

    public float[] predict(int[] u, int[] i) {
        if (u.length != i.length) throw new IllegalArgumentException("length mismatch");
        float[] out = new float[u.length];
        nativePredict(handle, u, i, out);
        return out;
    }

    /** The topN best items of each user, best first (ties: smaller item index); scores as predict() gives them. */
    public int[][] recommend(int[] users, int topN) {
        int[] items = new int[users.length * topN];
        float[] scores = new float[users.length * topN];
        nativeRecommend(handle, users, topN, items, scores);
        int[][] out = new int[users.length][];
        for (int a = 0; a < users.length; a++) out[a] = java.util.Arrays.copyOfRange(items, a * topN, (a + 1) * topN);
        return out;
    }

    public double rmse() {
        return nativeRmse(handle);
    }

    /** P (users x k) and Q (items x k), row-major. */
    public float[][] factors() {
        float[] p = new float[users * k], q = new float[items * k];
        nativeGetFactors(handle, p, q);
        return new float[][] {p, q};
    }

    public void setFactors(float[] p, float[] q) {
        if (p.length != users * k || q.length != items * k) throw new IllegalArgumentException("shape mismatch");
        nativeSetFactors(handle, p, q);
        initialised = true;
    }

    // ---- DSGD over the GPUs of one node (include/mfsgd.h, "DSGD driver"; INTEGRATION.md section 5) -------------

    /** 128 bytes naming a new ring; rank 0 calls it and hands the bytes to every rank (any transport the host has). */
    public static byte[] distributedId() {
        return nativeDsgdUniqueId();
    }

    /**
     * The global partitioner: {userBegin[nParts + 1], itemPart[items]} from the two degree arrays of ONE global rating
     * set -- users in contiguous ranges balanced by rating count (rank g keeps users [userBegin[g], userBegin[g+1])),
     * items in partitions balanced by rating count with the chain-critical items packed together.  A pure function:
     * every rank computes the same plan.
     */
    public static int[][] plan(long[] degUser, long[] degItem, int nParts) {
        int[] userBegin = new int[nParts + 1], itemPart = new int[degItem.length];
        nativeDsgdPlan(degUser, degItem, nParts, userBegin, itemPart);
        return new int[][] {userBegin, itemPart};
    }

    /**
     * This rank's share of {@code epochs} DSGD epochs: {@code u} are LOCAL user indices (global index - userOffset),
     * {@code i} global item indices; {@code itemPart} (nullable: i % nParts) is plan()'s item map; the factors are seeded
     * as the single-device run over usersTotal x items would seed them.  Collective: every rank calls it with the same
     * {@code id}, world and epochs.  Returns the GLOBAL RMSE after each epoch.  The first call builds the schedules and
     * joins the ring (ncclCommInitRank); later calls on the same ratings only train.
     */
    public double[] trainDistributed(int[] u, int[] i, float[] r, int epochs, int rank, int world, byte[] id, int[] itemPart,
                                     long userOffset, long usersTotal) {
        if (nParts < 2) throw new IllegalStateException("created without item partitions");
        if (u.length != i.length || u.length != r.length) throw new IllegalArgumentException("length mismatch");
        if (id == null || id.length != 128) throw new IllegalArgumentException("id must be the 128 bytes of distributedId()");
        if (ring == 0) {
            if (itemPart != null) nativeSetItemPartition(handle, itemPart);
            nativeSetRatings(handle, u, i, r);
            nativeInitPOffset(handle, seed, userOffset); // P row u at stream position (userOffset + u) * k
            ring = nativeDsgdCreate(handle, rank, world, id);
            slots = nParts / world;
            nativeDsgdInitQ(ring, seed, usersTotal);     // Q row i at stream position (usersTotal + i) * k
            initialised = true;
        }
        double[] rmse = new double[epochs];
        nativeDsgdTrain(ring, epochs, rmse);
        return rmse;
    }

    /** Global RMSE with the current factors (collective). */
    public double rmseDistributed() {
        return nativeDsgdRmse(ring);
    }

    /** This rank's P rows (users x k, row-major). */
    public float[] userFactors() {
        float[] p = new float[users * k];
        nativeGetUserFactors(handle, p);
        return p;
    }

    /**
     * The Q blocks this rank holds between epochs: for slot j, {@code partOut[j]} receives the partition id and the
     * returned array its rows x k block (rows in ascending item id within the partition).
     */
    public float[][] itemBlocks(int[] partOut) {
        float[][] out = new float[slots][];
        int[] partRows = new int[2];
        for (int j = 0; j < slots; j++) {
            nativeDsgdGetQ(ring, j, partRows, null);
            out[j] = new float[partRows[1] * k];
            nativeDsgdGetQ(ring, j, partRows, out[j]);
            if (partOut != null) partOut[j] = partRows[0];
        }
        return out;
    }

    @Override
    public void close() {
        if (ring != 0) {
            nativeDsgdDestroy(ring);
            ring = 0;
        }
        if (handle != 0) {
            nativeDestroy(handle);
            handle = 0;
        }
    }

    // Every native throws RuntimeException(mfsgd_last_error) on a non-zero status.
    private static native long nativeCreate(int users, int items, int k, float lr, float lambda, int device, int nParts);
    private static native void nativeDestroy(long h);
    private static native void nativeSetRatings(long h, int[] u, int[] i, float[] r);
    private static native void nativeInitFactors(long h, long seed);
    private static native void nativeSetFactors(long h, float[] p, float[] q);
    private static native void nativeGetFactors(long h, float[] p, float[] q);
    private static native void nativeTrain(long h, int epochs, double[] rmsePerEpoch);
    private static native double nativeRmse(long h);
    private static native void nativePredict(long h, int[] u, int[] i, float[] out);
    private static native void nativeRecommend(long h, int[] users, int topN, int[] items, float[] scores);
    // DSGD (mfsgd_dsgd_*, mfsgd_set_item_partition, mfsgd_init_p_offset)
    private static native byte[] nativeDsgdUniqueId();
    private static native void nativeDsgdPlan(long[] degUser, long[] degItem, int nParts, int[] userBegin, int[] itemPart);
    private static native void nativeSetItemPartition(long h, int[] itemPart);
    private static native void nativeInitPOffset(long h, long seed, long userOffset);
    private static native void nativeGetUserFactors(long h, float[] p);
    private static native long nativeDsgdCreate(long h, int rank, int world, byte[] id);
    private static native void nativeDsgdDestroy(long d);
    private static native void nativeDsgdInitQ(long d, long seed, long usersTotal);
    private static native void nativeDsgdTrain(long d, int epochs, double[] rmsePerEpoch);
    private static native double nativeDsgdRmse(long d);
    private static native void nativeDsgdGetQ(long d, int slot, int[] partRows, float[] block);
}
