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
    private final int users, items, k;
    private final long seed;
    private boolean initialised;

    public MatrixFactorizationSGD(int users, int items, int k, float lr, float lambda, long seed) {
        this.users = users;
        this.items = items;
        this.k = k;
        this.seed = seed;
        this.handle = nativeCreate(users, items, k, lr, lambda, /*device*/ 0);
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

    @Override
    public void close() {
        if (handle != 0) {
            nativeDestroy(handle);
            handle = 0;
        }
    }

    // Every native throws RuntimeException(mfsgd_last_error) on a non-zero status.
    private static native long nativeCreate(int users, int items, int k, float lr, float lambda, int device);
    private static native void nativeDestroy(long h);
    private static native void nativeSetRatings(long h, int[] u, int[] i, float[] r);
    private static native void nativeInitFactors(long h, long seed);
    private static native void nativeSetFactors(long h, float[] p, float[] q);
    private static native void nativeGetFactors(long h, float[] p, float[] q);
    private static native void nativeTrain(long h, int epochs, double[] rmsePerEpoch);
    private static native double nativeRmse(long h);
    private static native void nativePredict(long h, int[] u, int[] i, float[] out);
    private static native void nativeRecommend(long h, int[] users, int topN, int[] items, float[] scores);
}
