"""Streaming access to the embedding parquet files of the kNN stage: read only the row groups a base tile needs, one tile ahead of
the GPU.

The reference reads a whole file and slices it (``pq.read_table(...).slice(start, end)``, cu_knn.py:112-114) or streams record
batches with ``dataset.to_batches`` (cu_knn_ds.py:190) but searches them strictly one after the other.  At C2 size the parquet read
(0.5 s) is several times the GPU work (0.03-0.12 s), so the file path's wall clock is the read unless it overlaps:

* ``ParquetRows``  -- a lazy row window over one or more parquet files restricted to some columns: ``len``, ``schema``,
  ``slice(offset, length)`` (a ``pyarrow.Table`` of exactly those rows, built from the row groups that overlap them -- nothing else
  is read, nothing is concatenated: the table keeps one chunk per row-group piece, which the engine's upload pipeline gathers from
  directly) and ``iter_tables(batch_rows, start)``.
* ``Prefetch``     -- runs a generator on a background thread at most ``slots`` items ahead *including the one the consumer holds*:
  with two slots the reader decodes tile i+1 while tile i is on the GPU, and host memory never holds more than two tiles.

Nothing here touches the GPU; pyarrow releases the GIL while it decodes, and the engine's ctypes calls release it while they wait.
"""
from __future__ import annotations

import threading
from typing import Callable, Iterable, Iterator, List, Optional, Sequence

import pyarrow as pa
import pyarrow.parquet as pq


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
        for fi, path in enumerate(self.paths):
            md = pq.read_metadata(path)
            if schema is None:
                schema = pq.read_schema(path)
            for g in range(md.num_row_groups):
                n = md.row_group(g).num_rows
                if n > 0:
                    self._groups.append((fi, g, total, n))
                total += n
        self._file_rows = total
        self._count = total if count is None else max(0, min(int(count), total))
        self._full_schema = schema
        self._columns = list(columns) if columns is not None else list(schema.names)
        missing = [c for c in self._columns if c not in schema.names]
        if missing:
            raise KeyError(f"columns not in {self.paths[0]}: {missing[:5]}")
        self._tls = threading.local()          # ParquetFile handles, one set per reading thread

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
        return pa.schema([self._full_schema.field(c) for c in self._columns])

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
            f = cache[fi] = pq.ParquetFile(self.paths[fi])
        return f

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
