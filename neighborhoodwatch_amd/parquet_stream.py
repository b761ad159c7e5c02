"""Streaming access to the embedding parquet files of the kNN stage: read only the row groups a base tile needs, one tile ahead of
the GPU.

The reference reads a whole file and slices it (``pq.read_table(...).slice(start, end)``, cu_knn.py:112-114) or streams record
batches with ``dataset.to_batches`` (cu_knn_ds.py:190) but searches them strictly one after the other.  At C2 size the parquet read
(0.5 s) is several times the GPU work (0.03-0.12 s), so the file path's wall clock is the read unless it overlaps:

* ``ParquetRows``  -- a lazy row window over one or more parquet files restricted to some columns: ``len``, ``schema``,
  ``slice(offset, length)`` (a ``pyarrow.Table`` of exactly those rows, built from the row groups that overlap them -- nothing else
  is read, nothing is concatenated: the table keeps one chunk per row-group piece, which the engine's upload pipeline gathers from
  directly) and ``iter_tables(batch_rows, start)``.
* ``PieceReader``  -- decodes the row-group pieces of one or more row windows on SEVERAL reader threads (a ``ParquetFile`` handle each,
  memory-mapped) and hands them out in row order, a bounded number of pieces ahead of the consumer: the consumer appends piece g to the
  GPU's base (``KnnEngine.base_append_columns``: gather + H2D + layout) while pieces g+1.. are being decoded, so a one-tile job -- with
  ``mem_tune`` on 288 GB the whole file is ONE tile -- overlaps decode and upload too (round 4 decoded all 24.6 GB before the GPU saw a row).
* ``Prefetch``     -- runs a generator on a background thread at most ``slots`` items ahead *including the one the consumer holds*:
  with two slots the reader decodes tile i+1 while tile i is on the GPU, and host memory never holds more than two tiles.

Nothing here touches the GPU; pyarrow releases the GIL while it decodes, and the engine's ctypes calls release it while they wait.
"""
from __future__ import annotations

import os
import threading
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, List, Optional, Sequence, Tuple

import pyarrow as pa
import pyarrow.parquet as pq

from .hostinfo import size_arrow_pool


class ParquetRows:
    """Rows [0, count) of the concatenation of ``paths``, columns ``columns`` (all when None).  Only metadata is read here."""

    def __init__(self, paths, count: Optional[int] = None, columns: Optional[Sequence[str]] = None):
        self.paths: List[str] = [paths] if isinstance(paths, (str, bytes)) or hasattr(paths, "__fspath__") else list(paths)
        self.paths = [str(p) for p in self.paths]
        if not self.paths:
            raise ValueError("ParquetRows: no files")
        self._groups = []                      # (path index, row group, first row in the window's numbering, rows)
        total = 0
        schema = None
        self._metadata = []                    # the parsed footer of every file: 1536 columns x 40 row groups is megabytes of Thrift, parsed ONCE
        for fi, path in enumerate(self.paths):
            f = self._open(path)
            md = f.metadata
            self._metadata.append(md)
            if schema is None:
                schema = f.schema_arrow
            for g in range(md.num_row_groups):
                n = md.row_group(g).num_rows
                if n > 0:
                    self._groups.append((fi, g, total, n))
                total += n
        self._file_rows = total
        self._count = total if count is None else max(0, min(int(count), total))
        self._full_schema = schema
        self._columns = list(columns) if columns is not None else list(schema.names)
        have = set(schema.names)
        missing = [c for c in self._columns if c not in have]
        if missing:
            raise KeyError(f"columns not in {self.paths[0]}: {missing[:5]}")
        self._tls = threading.local()          # ParquetFile handles, one set per reading thread

    @staticmethod
    def _open(path, metadata=None) -> pq.ParquetFile:
        # memory-mapped: pyarrow decodes straight from the page cache instead of read()ing every column chunk into a buffer first --
        # 5.3 -> 14.5 GB/s of floats on the GPU box's host for the reference's layout (1536 float columns, 100 000-row groups;
        # tools/parquet_decode_probe.py, profiles/r05h_parquet_decode_probe.log), 3.5 -> 9.8 GB/s with pyarrow's default encoding
        try:
            return pq.ParquetFile(path, memory_map=True, metadata=metadata)
        except (OSError, pa.ArrowInvalid):
            return pq.ParquetFile(path, metadata=metadata)

    # ---- table-like face (what compute_knn / tune_memory / process_batches use of a pyarrow.Table) ----
    def __len__(self) -> int:
        return self._count

    @property
    def num_rows(self) -> int:
        return self._count

    @property
    def column_names(self) -> List[str]:
        return list(self._columns)

    @property
    def schema(self) -> pa.Schema:
        got = self.__dict__.get("_schema_of")                      # built once per selection: 1536 field lookups cost 0.6 ms, and callers ask per column
        if got is None or got[0] is not self._columns:
            got = (self._columns, pa.schema([self._full_schema.field(c) for c in self._columns]))
            self._schema_of = got
        return got[1]

    def select(self, columns: Sequence[str]) -> "ParquetRows":
        out = ParquetRows.__new__(ParquetRows)
        out.__dict__.update(self.__dict__)
        out._columns = list(columns)
        out._tls = threading.local()
        return out

    def head(self, count: int) -> "ParquetRows":
        out = self.select(self._columns)
        out._count = max(0, min(int(count), self._count))
        return out

    def _file(self, fi: int) -> pq.ParquetFile:
        cache = getattr(self._tls, "files", None)
        if cache is None:
            cache = self._tls.files = {}
        f = cache.get(fi)
        if f is None:
            f = cache[fi] = self._open(self.paths[fi], self._metadata[fi])       # a handle per reading thread, the footer parsed once for all
        return f

    def pieces(self, offset: int = 0, length: Optional[int] = None) -> List[Tuple[int, int, int, int, int]]:
        """The row-group pieces that make up rows [offset, offset + length) of the window, in row order:
        ``(file index, row group, first row of the piece in the window's numbering, rows to skip inside the group, rows)``."""
        offset = max(0, min(int(offset), self._count))
        end = self._count if length is None else max(offset, min(offset + int(length), self._count))
        out = []
        for fi, g, first, n in self._groups:
            if first + n <= offset:
                continue
            if first >= end:
                break
            lo = max(offset - first, 0)
            hi = min(end - first, n)
            if hi > lo:
                out.append((fi, g, first + lo, lo, hi - lo))
        return out

    def read_piece(self, piece: Tuple[int, int, int, int, int]) -> pa.Table:
        """One piece of ``pieces()`` as a table (decoded on the calling thread, with its own file handle)."""
        fi, g, _first, lo, n = piece
        t = self._file(fi).read_row_groups([g], columns=self._columns, use_threads=True)
        return t if (lo == 0 and n == len(t)) else t.slice(lo, n)

    def slice(self, offset: int = 0, length: Optional[int] = None) -> pa.Table:
        """Rows [offset, offset + length) of the window as a pyarrow.Table (pyarrow's ``Table.slice`` semantics: clipped at the end)."""
        offset = max(0, min(int(offset), self._count))
        end = self._count if length is None else max(offset, min(offset + int(length), self._count))
        pieces = []
        run_fi, run_groups, run_first = None, [], 0
        def flush_run():
            nonlocal run_groups
            if run_groups:
                t = self._file(run_fi).read_row_groups(run_groups, columns=self._columns, use_threads=True)
                lo = max(offset - run_first, 0)
                hi = min(end - run_first, len(t))
                pieces.append(t.slice(lo, hi - lo))
                run_groups = []
        for fi, g, first, n in self._groups:
            if first + n <= offset:
                continue
            if first >= end:
                break
            if run_groups and fi != run_fi:
                flush_run()
            if not run_groups:
                run_fi, run_first = fi, first
            run_groups.append(g)
        flush_run()
        if not pieces:
            return self.schema.empty_table()
        return pieces[0] if len(pieces) == 1 else pa.concat_tables(pieces)       # concat_tables keeps the chunks: no copy

    def provenance(self) -> str:
        """What the rows come from, for a resume checkpoint to be tied to: every file's path, size and modification time, the window's
        row count and columns (a stale checkpoint of another job with the same counts must not be folded into this one)."""
        import hashlib
        import os
        h = hashlib.sha1()
        for path in self.paths:
            st = os.stat(path)
            h.update(f"{os.path.abspath(path)}|{st.st_size}|{st.st_mtime_ns}\n".encode())
        h.update(f"rows={self._count}|cols={len(self._columns)}|{self._columns[0] if self._columns else ''}..{self._columns[-1] if self._columns else ''}".encode())
        return h.hexdigest()

    def to_table(self) -> pa.Table:
        return self.slice(0, self._count)

    def iter_tables(self, batch_rows: int, start: int = 0) -> Iterator[pa.Table]:
        """Tables of exactly ``batch_rows`` rows (the last one may be shorter) from row ``start`` on."""
        if batch_rows <= 0:
            raise ValueError("batch_rows must be positive")
        for off in range(int(start), self._count, int(batch_rows)):
            yield self.slice(off, batch_rows)


class TableRows:
    """The same face over rows that are already in memory (a pyarrow Table: the materialising fallback of cu_knn_ds.slice_dataset, or a
    dataset without files behind it)."""

    def __init__(self, table: pa.Table):
        self._t = table

    def __len__(self) -> int:
        return len(self._t)

    num_rows = property(__len__)

    @property
    def column_names(self) -> List[str]:
        return list(self._t.column_names)

    @property
    def schema(self) -> pa.Schema:
        return self._t.schema

    def select(self, columns: Sequence[str]) -> "TableRows":
        return TableRows(self._t.select(list(columns)))

    def slice(self, offset: int = 0, length: Optional[int] = None) -> pa.Table:
        return self._t.slice(offset, length)

    def to_table(self) -> pa.Table:
        return self._t

    def provenance(self) -> str:
        """In-memory rows have no file identity: a checkpoint written over them is only good inside this process."""
        import os
        return f"memory:{os.getpid()}:{id(self._t)}:{len(self._t)}x{self._t.num_columns}"


def default_readers() -> int:
    """Reader threads of a PieceReader: 4 unless NWKNN_READERS says otherwise (on the GPU box's host 2-4 readers with pyarrow's column threads
    inside each decode fastest -- 16-18 GB/s of floats; more readers fight for the same column-thread pool)."""
    try:
        return max(1, int(os.environ.get("NWKNN_READERS", "4")))
    except ValueError:
        return 4


class PieceReader:
    """Decode ``pieces`` (each ``(rows: ParquetRows, piece)``) on ``readers`` threads and iterate the tables IN ORDER, at most ``ahead``
    pieces decoded or in flight beyond the one the consumer holds.  An exception in a reader surfaces at the position of its piece."""

    def __init__(self, pieces: Sequence[Tuple["ParquetRows", tuple]], readers: Optional[int] = None, ahead: Optional[int] = None,
                 ahead_bytes: int = 6 << 30, transform: Optional[Callable] = None):
        self._pieces = list(pieces)
        self._transform = transform              # applied to every decoded table ON ITS READER THREAD (the consumer gets the result)
        self._readers = readers or default_readers()
        size_arrow_pool()                        # pyarrow's column threads: as many as the container may run, not as the machine shows
        if ahead is None:
            # bounded by host memory, not by count: a piece of the reference's published run is 100 000 rows x 1536 floats = 614 MB
            biggest = 1
            for rows, pc in self._pieces[:64]:
                row_bytes = sum((f.type.bit_width // 8) if pa.types.is_floating(f.type) else
                                (f.type.list_size * f.type.value_type.bit_width // 8 if pa.types.is_fixed_size_list(f.type) else 8) for f in rows.schema)
                biggest = max(biggest, pc[4] * max(row_bytes, 1))
            ahead = max(self._readers, min(16, ahead_bytes // biggest))
        self._ahead = max(1, int(ahead))
        self._pool = ThreadPoolExecutor(max_workers=self._readers, thread_name_prefix="nw-reader")
        self._queue: deque = deque()
        self._next = 0
        self.max_inflight = 0
        self._fill()

    def _fill(self):
        while self._next < len(self._pieces) and len(self._queue) < self._ahead:
            rows, pc = self._pieces[self._next]
            self._queue.append(self._pool.submit(self._read, rows, pc))
            self._next += 1
        self.max_inflight = max(self.max_inflight, len(self._queue))

    def _read(self, rows, pc):
        t = rows.read_piece(pc)
        return t if self._transform is None else self._transform(t)

    def __iter__(self):
        return self

    def __next__(self):
        if not self._queue:
            raise StopIteration
        fut = self._queue.popleft()
        self._fill()                                             # keep the readers busy while the consumer works on this piece
        return fut.result()

    def close(self):
        for f in self._queue:
            f.cancel()
        self._queue.clear()
        self._pool.shutdown(wait=False)                          # idle readers wind down (and unmap their handles) behind the consumer's back

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Prefetch:
    """Iterate ``make_iter()`` on a background thread, at most ``slots`` items alive: the ones queued plus the one the consumer is
    holding (it counts as released when the consumer asks for the next).  An exception in the producer is re-raised in the
    consumer at the point where the failed item would have arrived; ``close()`` stops the producer early."""

    _END = object()

    def __init__(self, make_iter: Callable[[], Iterable], slots: int = 2):
        if slots < 1:
            raise ValueError("slots must be >= 1")
        self._make_iter = make_iter
        self._slots = threading.Semaphore(slots)
        self._items: list = []
        self._cv = threading.Condition()
        self._stop = False
        self._holding = False
        self.max_alive = 0                     # high-water mark of items alive at once (tests read it)
        self._alive = 0
        self._thread = threading.Thread(target=self._produce, name="nw-prefetch", daemon=True)
        self._thread.start()

    def _produce(self):
        try:
            it = iter(self._make_iter())
            while True:
                self._slots.acquire()                            # room for one more item before it is read
                if self._stop:
                    break
                try:
                    item = next(it)
                except StopIteration:
                    break
                with self._cv:
                    self._alive += 1
                    self.max_alive = max(self.max_alive, self._alive)
                    self._items.append((item, None))
                    self._cv.notify_all()
        except BaseException as e:                               # noqa: BLE001 -- handed to the consumer
            with self._cv:
                self._items.append((None, e))
                self._cv.notify_all()
            return
        with self._cv:
            self._items.append((Prefetch._END, None))
            self._cv.notify_all()

    def __iter__(self):
        return self

    def __next__(self):
        if self._holding:                                        # the consumer is done with the previous item
            self._holding = False
            with self._cv:
                self._alive -= 1
            self._slots.release()
        with self._cv:
            while not self._items:
                self._cv.wait()
            item, err = self._items.pop(0)
        if err is not None:
            raise err
        if item is Prefetch._END:
            raise StopIteration
        self._holding = True
        return item

    def close(self):
        self._stop = True
        self._slots.release()
        self._thread.join(timeout=30)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
