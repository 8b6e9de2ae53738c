// sqlite_ingest.cpp -- comparison rows from the result matrices straight into the run database.
//
// "Next" row 1 of SURVEY.md section 8(f).  In the reference every comparison travels as a Python dict
// through a JSON file, is parsed back by the parent process and reaches SQLite through SQLAlchemy's
// INSERT OR IGNORE (pyani_plus/private_cli.py:507-614, pyani_plus/db_orm.py:1076).  Here the rows are bound
// from the N x N matrices and stepped through one prepared statement: no per-row Python objects, which is
// what the executemany form of rundb.ingest_matrices spends most of its two microseconds per row on.
//
// The image carries libsqlite3.so.0 (the library Python's own sqlite3 module links) but not its header, so
// the dozen entry points used here are declared by hand -- the public, stable C API of SQLite 3 -- and
// resolved with dlopen at the first call.  Without the library the call fails with PA_E_IO and
// rundb falls back on Python's sqlite3 module (same statement, same rows).
#include <dlfcn.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>

#include "../../include/pyani_hip.h"

void pa_set_error(const char *fmt, ...);

namespace {

struct sqlite3;
struct sqlite3_stmt;
constexpr int kSqliteOk = 0, kSqliteDone = 101, kOpenReadWrite = 0x2;
using destructor_t = void (*)(void *);

struct Api {
  void *handle = nullptr;
  int (*open_v2)(const char *, sqlite3 **, int, const char *) = nullptr;
  int (*close)(sqlite3 *) = nullptr;
  int (*exec)(sqlite3 *, const char *, int (*)(void *, int, char **, char **), void *, char **) = nullptr;
  int (*prepare_v2)(sqlite3 *, const char *, int, sqlite3_stmt **, const char **) = nullptr;
  int (*bind_text)(sqlite3_stmt *, int, const char *, int, destructor_t) = nullptr;
  int (*bind_int64)(sqlite3_stmt *, int, long long) = nullptr;
  int (*bind_double)(sqlite3_stmt *, int, double) = nullptr;
  int (*bind_null)(sqlite3_stmt *, int) = nullptr;
  int (*step)(sqlite3_stmt *) = nullptr;
  int (*reset)(sqlite3_stmt *) = nullptr;
  int (*finalize)(sqlite3_stmt *) = nullptr;
  int (*busy_timeout)(sqlite3 *, int) = nullptr;
  int (*total_changes)(sqlite3 *) = nullptr;
  const char *(*errmsg)(sqlite3 *) = nullptr;
  bool ok = false;
};

const Api &api() {
  static Api a;
  static std::once_flag once;
  std::call_once(once, [] {
    a.handle = dlopen("libsqlite3.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!a.handle) return;
    bool all = true;
    auto sym = [&](auto &fn, const char *name) {
      fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(a.handle, name));
      all = all && fn != nullptr;
    };
    sym(a.open_v2, "sqlite3_open_v2");
    sym(a.close, "sqlite3_close");
    sym(a.exec, "sqlite3_exec");
    sym(a.prepare_v2, "sqlite3_prepare_v2");
    sym(a.bind_text, "sqlite3_bind_text");
    sym(a.bind_int64, "sqlite3_bind_int64");
    sym(a.bind_double, "sqlite3_bind_double");
    sym(a.bind_null, "sqlite3_bind_null");
    sym(a.step, "sqlite3_step");
    sym(a.reset, "sqlite3_reset");
    sym(a.finalize, "sqlite3_finalize");
    sym(a.busy_timeout, "sqlite3_busy_timeout");
    sym(a.total_changes, "sqlite3_total_changes");
    sym(a.errmsg, "sqlite3_errmsg");
    a.ok = all;
  });
  return a;
}

constexpr const char *kInsert =
    "INSERT OR IGNORE INTO comparisons (query_hash, subject_hash, configuration_id, identity, aln_length, "
    "sim_errors, cov_query, uname_system, uname_release, uname_machine) VALUES (?,?,?,?,?,?,?,?,?,?)";

}  // namespace

extern "C" int pa_sqlite_insert_comparisons(const char *database, int64_t configuration_id, const char *uname_system,
                                            const char *uname_release, const char *uname_machine,
                                            const char *const *query_hashes, uint32_t n_queries,
                                            const char *const *subject_hashes, uint32_t n_subjects,
                                            const double *identity, const double *cov_query, const uint8_t *is_null,
                                            uint64_t *rows_inserted) {
  return pa_sqlite_insert_comparisons_ex(database, configuration_id, uname_system, uname_release, uname_machine, query_hashes,
                                         n_queries, subject_hashes, n_subjects, identity, cov_query, is_null, nullptr, nullptr,
                                         rows_inserted);
}

extern "C" int pa_sqlite_insert_comparisons_ex(const char *database, int64_t configuration_id, const char *uname_system,
                                               const char *uname_release, const char *uname_machine,
                                               const char *const *query_hashes, uint32_t n_queries,
                                               const char *const *subject_hashes, uint32_t n_subjects,
                                               const double *identity, const double *cov_query, const uint8_t *is_null,
                                               const int64_t *aln_length, const int64_t *sim_errors,
                                               uint64_t *rows_inserted) {
  if (rows_inserted) *rows_inserted = 0;
  if ((aln_length == nullptr) != (sim_errors == nullptr)) {
    pa_set_error("pa_sqlite_insert_comparisons_ex: aln_length and sim_errors go together");
    return PA_E_INVALID;
  }
  if (!database || !uname_system || !uname_release || !uname_machine || (n_queries && !query_hashes) ||
      (n_subjects && !subject_hashes) || ((uint64_t)n_queries * n_subjects && (!identity || !cov_query || !is_null))) {
    pa_set_error("pa_sqlite_insert_comparisons: null argument");
    return PA_E_INVALID;
  }
  const Api &a = api();
  if (!a.ok) {
    const char *why = a.handle ? "missing symbol" : dlerror();  // dlerror() may hold nothing by now
    pa_set_error("pa_sqlite_insert_comparisons: libsqlite3.so.0 could not be loaded (%s)", why ? why : "no loader message");
    return PA_E_IO;
  }
  sqlite3 *db = nullptr;
  if (a.open_v2(database, &db, kOpenReadWrite, nullptr) != kSqliteOk) {
    pa_set_error("pa_sqlite_insert_comparisons: cannot open %s: %s", database, db ? a.errmsg(db) : "out of memory");
    if (db) a.close(db);
    return PA_E_IO;
  }
  auto fail = [&](const char *what) {
    pa_set_error("pa_sqlite_insert_comparisons: %s: %s", what, a.errmsg(db));
    a.exec(db, "ROLLBACK", nullptr, nullptr, nullptr);
    a.close(db);
    return PA_E_IO;
  };
  a.busy_timeout(db, 30000);
  // synchronous=NORMAL (not OFF): the database is the user's persistent multi-run file, and with OFF a crash of the
  // machine during the commit can corrupt all of it, other runs' rows included.  The insert is one transaction, so
  // NORMAL costs a handful of fsyncs at commit time.  PA_SQLITE_SYNCHRONOUS=OFF opts into the faster, riskier form.
  const char *sync_env = getenv("PA_SQLITE_SYNCHRONOUS");
  const bool sync_off = sync_env && (sync_env[0] == 'O' || sync_env[0] == 'o' || sync_env[0] == '0');
  if (a.exec(db, sync_off ? "PRAGMA synchronous=OFF; PRAGMA cache_size=-1048576; BEGIN IMMEDIATE"
                          : "PRAGMA synchronous=NORMAL; PRAGMA cache_size=-1048576; BEGIN IMMEDIATE",
             nullptr, nullptr, nullptr) != kSqliteOk)
    return fail("begin");
  // Rows go in kBlock at a time through one statement with kBlock value tuples (fewer trips through
  // sqlite3_step / sqlite3_reset and the statement's set-up and tear-down code); the last rows of a matrix row
  // through the one-tuple statement.  Constant columns are bound once: bindings survive sqlite3_reset.
  constexpr uint32_t kBlock = 64, kCols = 10;  // 640 parameters: below the 999 older SQLite builds allow
  std::string many = "INSERT OR IGNORE INTO comparisons (query_hash, subject_hash, configuration_id, identity, aln_length, "
                     "sim_errors, cov_query, uname_system, uname_release, uname_machine) VALUES ";
  for (uint32_t r = 0; r < kBlock; ++r) many += r ? ",(?,?,?,?,?,?,?,?,?,?)" : "(?,?,?,?,?,?,?,?,?,?)";
  sqlite3_stmt *st = nullptr, *stm = nullptr;
  if (a.prepare_v2(db, kInsert, -1, &st, nullptr) != kSqliteOk) return fail("prepare");
  if (a.prepare_v2(db, many.c_str(), -1, &stm, nullptr) != kSqliteOk) { a.finalize(st); return fail("prepare (block form)"); }
  const destructor_t kStatic = nullptr;  // SQLITE_STATIC: the strings outlive the statements
  const int before = a.total_changes(db);
  bool bad = false;
  auto bind_constants = [&](sqlite3_stmt *s, uint32_t tuples) {
    for (uint32_t r = 0; r < tuples; ++r) {
      const int o = (int)(r * kCols);
      bad |= a.bind_int64(s, o + 3, configuration_id) != kSqliteOk;
      bad |= a.bind_null(s, o + 5) != kSqliteOk;  // aln_length and sim_errors: never set by the sourmash method
      bad |= a.bind_null(s, o + 6) != kSqliteOk;  // (pyani_plus/private_cli.py:1866-1880); bound per row for fastANI's
      bad |= a.bind_text(s, o + 8, uname_system, -1, kStatic) != kSqliteOk;
      bad |= a.bind_text(s, o + 9, uname_release, -1, kStatic) != kSqliteOk;
      bad |= a.bind_text(s, o + 10, uname_machine, -1, kStatic) != kSqliteOk;
    }
  };
  bind_constants(st, 1);
  bind_constants(stm, kBlock);
  auto bind_pair = [&](sqlite3_stmt *s, int o, uint32_t sub, uint64_t cell) {
    bad |= a.bind_text(s, o + 2, subject_hashes[sub], -1, kStatic) != kSqliteOk;
    if (is_null[cell]) {
      bad |= a.bind_null(s, o + 4) != kSqliteOk;
      bad |= a.bind_null(s, o + 7) != kSqliteOk;
      if (aln_length) {
        bad |= a.bind_null(s, o + 5) != kSqliteOk;
        bad |= a.bind_null(s, o + 6) != kSqliteOk;
      }
    } else {
      bad |= a.bind_double(s, o + 4, identity[cell]) != kSqliteOk;
      bad |= a.bind_double(s, o + 7, cov_query[cell]) != kSqliteOk;
      if (aln_length) {  // fastANI's proxy columns (pyani_plus/private_cli.py:1072-1080)
        bad |= a.bind_int64(s, o + 5, aln_length[cell]) != kSqliteOk;
        bad |= a.bind_int64(s, o + 6, sim_errors[cell]) != kSqliteOk;
      }
    }
  };
  uint64_t stepped = 0;
  for (uint32_t q = 0; q < n_queries && !bad; ++q) {
    const uint64_t row = (uint64_t)q * n_subjects;
    bad |= a.bind_text(st, 1, query_hashes[q], -1, kStatic) != kSqliteOk;
    for (uint32_t r = 0; r < kBlock; ++r) bad |= a.bind_text(stm, (int)(r * kCols) + 1, query_hashes[q], -1, kStatic) != kSqliteOk;
    uint32_t s = 0;
    for (; s + kBlock <= n_subjects && !bad; s += kBlock) {
      for (uint32_t r = 0; r < kBlock; ++r) bind_pair(stm, (int)(r * kCols), s + r, row + s + r);
      bad |= a.step(stm) != kSqliteDone;
      a.reset(stm);
      stepped += kBlock;
    }
    for (; s < n_subjects && !bad; ++s) {
      bind_pair(st, 0, s, row + s);
      bad |= a.step(st) != kSqliteDone;
      a.reset(st);
      ++stepped;
    }
  }
  a.finalize(stm);
  if (bad) {
    a.finalize(st);
    char what[64];
    snprintf(what, sizeof(what), "insert near row %llu", (unsigned long long)stepped);
    return fail(what);
  }
  a.finalize(st);
  if (a.exec(db, "COMMIT", nullptr, nullptr, nullptr) != kSqliteOk) return fail("commit");
  if (rows_inserted) *rows_inserted = (uint64_t)(uint32_t)(a.total_changes(db) - before);
  a.close(db);
  return PA_OK;
}
