/*
 * HipBVGraph.java — the class a maintainer of webgraph-big adds (package it.unimi.dsi.big.webgraph) to route BVGraph's decode path to
 * libbvgraph_hip.so (include/bvgraph_hip.h).  Complete source; NOT compiled in this repository: the image has no JDK (no javac, no jni.h),
 * and the classes it extends (ImmutableGraph, NodeIterator, LazyLongIterators) and fastutil live in the reference, which is not vendored.
 *
 * What it replaces, line by line:  load* -> BVGraph.java:1345-1574;  outdegree -> :821-851;  successors / successorBigArray -> :860-867, 995-1097;
 * nodeIterator -> :1100-1265 (BVGraphNodeIterator);  copy -> :553-578;  splitNodeIterators -> ImmutableGraph.java:405-436.
 * Status -> exception: bvgraph_hip_jni.c throw_status() (BVG:823,863,1000,1128 IllegalArgumentException; :701,832,1136 IllegalStateException;
 * :631,864 UnsupportedOperationException; :1492-1497 IOException).
 */
package it.unimi.dsi.big.webgraph;

import it.unimi.dsi.fastutil.longs.LongBigArrays;
import it.unimi.dsi.logging.ProgressLogger;

import java.io.IOException;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;
import java.nio.LongBuffer;
import java.util.NoSuchElementException;

public final class HipBVGraph extends ImmutableGraph implements AutoCloseable {
	static { System.loadLibrary("bvgraph_hip_jni"); }                    // bvgraph_hip_jni.c, linked against libbvgraph_hip.so

	/** Nodes decoded per native call by a node iterator (one kernel launch + one PCIe transfer each). */
	public static final int BATCH_NODES = 1 << 16;

	private long handle;                                                 // bvg_graph*
	private final CharSequence basename;
	private final long n, m;
	private final int device;

	// ---- natives: one bvg_* call each; a negative status is thrown by the glue as the mapped exception (no partial results) ----
	private static native long nOpen(String basename, int loadMode, int device) throws IOException;                 // bvg_open
	private static native long nCopy(long h);                                                                      // bvg_copy
	private static native void nClose(long h);                                                                     // bvg_close
	private static native long[] nInfo(long h);                                                                    // bvg_info -> {nodes, arcs, W, maxRef, minInterval, zetaK}
	private static native void nOutdegrees(long h, long from, long to, int[] out);                                 // bvg_outdegrees
	/** bvg_decode_range into page-locked direct buffers: returns the number of successors written, or -(needed) when succ is too small (nothing written). */
	private static native long nDecodeRange(long h, long from, long to, IntBuffer outdeg, LongBuffer succ);
	private static native long nSuccessorsBatch(long h, long[] nodes, IntBuffer outdeg, LongBuffer succ);          // bvg_successors_batch (a BFS / HyperBall frontier in one launch)
	private static native ByteBuffer nHostAlloc(long bytes);                                                       // bvg_host_alloc -> NewDirectByteBuffer
	private static native void nHostFree(ByteBuffer b);                                                            // bvg_host_free
	private static native long[] nScan(long h, long from, long to);                                                // bvg_scan -> {nodes, arcs, chk, graph bytes, index bytes, kernel ns}
	private static native long[] nScanMulti(long[] handles, int balance);                                          // bvg_scan_multi (one handle per device) -> {nodes, arcs, chk, kernel ns of the slowest shard}
	private static native long[] nShardBounds(long h, int k, int balance);                                         // bvg_shard_bounds -> k + 1 node ids
	private static native long[] nBuildIndex(long h, long from, long to);                                          // bvg_build_index -> {entries, bytes}
	private static native void nSaveIndex(long h, String path) throws IOException;                                 // bvg_save_index
	private static native void nLoadIndex(long h, String path) throws IOException;                                 // bvg_load_index
	private static native void nSetIndexMode(long h, int mode);                                                    // bvg_set_tuning {no_index}: 0 full index, 1 none, 2 marks only

	private HipBVGraph(final long h, final CharSequence basename, final int device) {
		this.handle = h; this.basename = basename; this.device = device;
		final long[] i = nInfo(h); n = i[0]; m = i[1];
	}

	// ---- the entry points ImmutableGraph.load*(basename) finds by reflection (ImmutableGraph.java:104-121, 674-713; BVGraph.java:1345-1464) ----
	public static HipBVGraph load(final CharSequence basename) throws IOException { return load(basename, null); }
	public static HipBVGraph load(final CharSequence basename, final ProgressLogger pl) throws IOException { return new HipBVGraph(nOpen(basename.toString(), 1, 0), basename, 0); }
	public static HipBVGraph loadMapped(final CharSequence basename, final ProgressLogger pl) throws IOException { return new HipBVGraph(nOpen(basename.toString(), 2, 0), basename, 0); }
	public static HipBVGraph loadMapped(final CharSequence basename) throws IOException { return loadMapped(basename, null); }
	public static HipBVGraph loadOffline(final CharSequence basename, final ProgressLogger pl) throws IOException { return new HipBVGraph(nOpen(basename.toString(), -1, 0), basename, 0); }
	public static HipBVGraph loadOffline(final CharSequence basename) throws IOException { return loadOffline(basename, null); }
	@Deprecated public static HipBVGraph loadSequential(final CharSequence basename, final ProgressLogger pl) throws IOException { return new HipBVGraph(nOpen(basename.toString(), 0, 0), basename, 0); }
	@Deprecated public static HipBVGraph loadSequential(final CharSequence basename) throws IOException { return loadSequential(basename, null); }
	/** The same graph on another device of the node (one handle per GPU: scanMulti). */
	public static HipBVGraph load(final CharSequence basename, final int device) throws IOException { return new HipBVGraph(nOpen(basename.toString(), 1, device), basename, device); }

	@Override public long numNodes() { return n; }
	@Override public long numArcs() { if (m < 0) throw new UnsupportedOperationException(); return m; }
	@Override public boolean randomAccess() { return true; }
	@Override public boolean hasCopiableIterators() { return true; }
	@Override public CharSequence basename() { return basename; }
	/** A flyweight: shares the stream, offsets and index in HBM, owns its stream and workspaces; usable from another thread (BVGraph.java:553-578, ImmutableGraph.java:187-197). */
	@Override public HipBVGraph copy() { ensureOpen(); return new HipBVGraph(nCopy(handle), basename, device); }
	@Override public void close() { if (handle != 0) { nClose(handle); handle = 0; } }
	@Override @SuppressWarnings("deprecation") protected void finalize() throws Throwable { try { close(); } finally { super.finalize(); } }   // as the reference closes its stream, BVGraph.java:1211-1220
	private void ensureOpen() { if (handle == 0) throw new IllegalStateException("This graph has been closed"); }

	@Override public long outdegree(final long x) {                                                  // BVGraph.java:821-842
		ensureOpen();
		if (x < 0 || x >= n) throw new IllegalArgumentException("Node index out of range: " + x);
		final int[] d = new int[1]; nOutdegrees(handle, x, x + 1, d); return d[0];
	}

	/** Random access: one native call per node (the reference recurses through the reference chain, BVGraph.java:1084; the library decodes the chain as the block's halo). */
	@Override public long[][] successorBigArray(final long x) {                                      // BVGraph.java:860-867
		ensureOpen();
		if (x < 0 || x >= n) throw new IllegalArgumentException("Node index out of range: " + x);
		try (Batch b = new Batch(1, 1024)) {
			b.decode(x, x + 1);
			final long[][] a = LongBigArrays.newBigArray(b.deg.get(0));
			b.copyList(0, a);
			return a;
		}
	}
	@Override public LazyLongIterator successors(final long x) { final long[][] a = successorBigArray(x); return LazyLongIterators.wrap(a, LongBigArrays.length(a)); }

	/** A whole frontier (BFS, HyperBall) in ONE launch: successors of nodes[i] are succ[cum[i] .. cum[i + 1]); returns {outdegrees, successors}. */
	public long[][] successorsOf(final long[] nodes) {
		ensureOpen();
		try (Batch b = new Batch(nodes.length, Math.max(1024, 16L * nodes.length))) {
			long got;
			while ((got = nSuccessorsBatch(handle, nodes, b.deg, b.succ)) < 0) b.growSucc(-got);
			final long[] d = new long[nodes.length]; for (int i = 0; i < nodes.length; i++) d[i] = b.deg.get(i);
			final long[] s = new long[(int)got]; b.succ.position(0); b.succ.get(s, 0, (int)got);
			return new long[][] { d, s };
		}
	}

	/** Page-locked buffers of one batch: bvg_decode_range writes them at the PCIe rate (a Java-heap long[] has to be staged by the runtime). */
	private final class Batch implements AutoCloseable {
		ByteBuffer degMem, succMem; IntBuffer deg; LongBuffer succ; long[] cum;
		Batch(final int nodes, final long succCap) {
			degMem = nHostAlloc(4L * nodes); deg = degMem.order(ByteOrder.nativeOrder()).asIntBuffer();
			succMem = nHostAlloc(8L * succCap); succ = succMem.order(ByteOrder.nativeOrder()).asLongBuffer();
			cum = new long[nodes + 1];
		}
		/** The new block first: if its allocation throws, the old one is still the one close() frees (exactly once). */
		void growSucc(final long needed) { final ByteBuffer m = nHostAlloc(8L * needed); nHostFree(succMem); succMem = m; succ = succMem.order(ByteOrder.nativeOrder()).asLongBuffer(); }
		void decode(final long from, final long to) {
			long got;
			while ((got = nDecodeRange(handle, from, to, deg, succ)) < 0) growSucc(-got);             // the size the library asked for (BVG_E_CAPACITY): nothing was written
			final int k = (int)(to - from);
			for (int i = 0; i < k; i++) cum[i + 1] = cum[i] + deg.get(i);
		}
		/** Successors of node i of the batch into the first deg(i) elements of a big array (one bulk copy out of the pinned buffer per 2^27-element segment). */
		void copyList(final int i, final long[][] dst) {
			long p = cum[i]; final long e = cum[i + 1];
			for (int seg = 0; p < e; seg++) {
				final int len = (int)Math.min(e - p, dst[seg].length);
				succ.position((int)p); succ.get(dst[seg], 0, len);                                    // (a batch holds < 2^31 successors: the native side is asked for BATCH_NODES nodes at a time)
				p += len;
			}
		}
		@Override public void close() { if (degMem != null) { nHostFree(degMem); nHostFree(succMem); degMem = succMem = null; } }
	}

	/** BVGraph.BVGraphNodeIterator (BVGraph.java:1100-1245) fed by batched native decodes. */
	@Override public NodeIterator nodeIterator(final long from) { ensureOpen(); return new BatchIterator(this, from, Long.MAX_VALUE); }
	@Override public NodeIterator nodeIterator() { return nodeIterator(0); }

	private static final class BatchIterator extends NodeIterator implements AutoCloseable {
		private final HipBVGraph g; private final long from, limit;
		private long curr, b0, b1;                                       // current node; the batch holds nodes [b0, b1)
		private Batch batch;
		/** The array successorBigArray() hands out: owned by the iterator, reused from node to node, valid until the next nextLong(), possibly longer than the
		 *  outdegree (NodeIterator.java:80-96) -- exactly what the reference's iterator does with the slot of its cyclic window (BVGraph.java:1192-1203).  No allocation
		 *  per node: one bulk copy out of the pinned batch buffer, where the reference's iterator decodes into its array element by element. */
		private long[][] list = LongBigArrays.newBigArray(1024);
		private boolean listValid;

		BatchIterator(final HipBVGraph g, final long from, final long upperBound) {
			if (from < 0 || from > g.n) throw new IllegalArgumentException("Node index out of range: " + from);    // BVGraph.java:1128
			this.g = g; this.from = from; curr = from - 1; limit = Math.min(upperBound, g.n) - 1; b0 = b1 = from;
		}
		@Override public boolean hasNext() { return curr < limit; }                                    // BVGraph.java:1179-1181
		@Override public long nextLong() {                                                             // BVGraph.java:1164-1176
			if (!hasNext()) throw new NoSuchElementException();
			if (++curr >= b1) {
				if (batch == null) batch = g.new Batch(BATCH_NODES, 16L * BATCH_NODES);
				b0 = curr; b1 = Math.min(curr + BATCH_NODES, limit + 1);
				batch.decode(b0, b1);
			}
			listValid = false;
			return curr;
		}
		@Override public long outdegree() { if (curr == from - 1) throw new IllegalStateException(); return batch.deg.get((int)(curr - b0)); }   // BVGraph.java:1206-1209
		@Override public long[][] successorBigArray() {                                                // BVGraph.java:1192-1203
			if (curr == from - 1) throw new IllegalStateException();
			if (!listValid) {
				final long d = outdegree();
				if (LongBigArrays.length(list) < d) list = LongBigArrays.newBigArray(Math.max(d, 2 * LongBigArrays.length(list)));
				batch.copyList((int)(curr - b0), list);
				listValid = true;
			}
			return list;
		}
		@Override public LazyLongIterator successors() { return LazyLongIterators.wrap(successorBigArray(), outdegree()); }   // BVGraph.java:1184-1189
		/** An iterator over [curr + 1, upperBound) for another thread: its own flyweight handle, its own buffers (NodeIterator.java:98-111, BVGraph.java:1223-1229). */
		@Override public NodeIterator copy(final long upperBound) { return new BatchIterator(g.copy(), curr + 1, upperBound); }
		@Override public void close() { if (batch != null) { batch.close(); batch = null; } }
		@Override @SuppressWarnings("deprecation") protected void finalize() throws Throwable { try { close(); } finally { super.finalize(); } }
	}

	/** ImmutableGraph.splitNodeIterators (ImmutableGraph.java:405-436) with the library's arc-balanced bounds instead of ceil(n / k) nodes each. */
	@Override public NodeIterator[] splitNodeIterators(final int howMany) {
		ensureOpen();
		if (numNodes() == 0 && howMany == 0) return new NodeIterator[0];
		if (howMany < 1) throw new IllegalArgumentException();
		final long[] b = nShardBounds(handle, howMany, 2 /* BVG_BALANCE_ARCS */);
		final NodeIterator[] it = new NodeIterator[howMany];
		for (int i = 0; i < howMany; i++) it[i] = b[i] < b[i + 1] ? new BatchIterator(copy(), b[i], b[i + 1]) : NodeIterator.EMPTY;
		return it;
	}

	// ---- beyond the reference's API: the fused on-chip scan (what test/SpeedTest.java:127-141 measures, without moving a successor to the host) ----
	/** {nodes, arcs, checksum, graph bytes, index bytes, kernel ns} of a scan of [from, to): successors are decoded, counted and folded into the checksum of include/bvgraph_hip.h on the GPU. */
	public long[] scan(final long from, final long to) { ensureOpen(); return nScan(handle, from, to); }
	/** One graph per device (load(basename, device)): shard i of the arc-balanced split runs on handles[i] from its own host thread; the three words are summed on the host. */
	public static long[] scanMulti(final HipBVGraph[] perDevice) {
		final long[] h = new long[perDevice.length]; for (int i = 0; i < h.length; i++) { perDevice[i].ensureOpen(); h[i] = perDevice[i].handle; }
		return nScanMulti(h, 2 /* BVG_BALANCE_ARCS */);
	}
	/** Builds (and validates) the residual skip index of [from, to) now instead of in the first scan; returns {entries, bytes}. */
	public long[] buildIndex(final long from, final long to) { ensureOpen(); return nBuildIndex(handle, from, to); }
	public void saveIndex(final CharSequence path) throws IOException { ensureOpen(); nSaveIndex(handle, path.toString()); }      // basename + ".bvgidx": bvg_open picks it up (cf. the .obl file, BVGraph.java:1545-1555)
	public void loadIndex(final CharSequence path) throws IOException { ensureOpen(); nLoadIndex(handle, path.toString()); }
	/** A handle that scans a graph once should neither build nor read the index. */
	public void setNoIndex(final boolean noIndex) { ensureOpen(); nSetIndexMode(handle, noIndex ? 1 : 0); }
	/** How much index the scans of this handle build and use: {@code 0} the full residual skip index (~50 % of the stream in HBM), {@code 1} none (a graph scanned once),
	 *  {@code 2} marks only (round 6: one byte per block + entries for lists of &ge; 4 096 residuals, ~0.03 % of the stream; ~36 % of the full index's rate on a dense web graph, 94 % on cnr-2000). */
	public void setIndexMode(final int mode) { ensureOpen(); nSetIndexMode(handle, mode); }
}
