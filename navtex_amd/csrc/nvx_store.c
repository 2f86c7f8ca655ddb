/* nvx_store.c -- SQLite message sink, compatible with the database the
 * reference's web server reads.
 *
 * What it mirrors:
 *   schema        receiver/generate_db.sql:3-8   (tables messages, config + 4 default rows)
 *   add_message   receiver/message_store.c:59-97 (delete the old copy of bbbb, insert the
 *                 new one with a UTC "%Y-%m-%d %H:%M" timestamp and age 'NEW';
 *                 returns 0 / -1 (open) / -2 (insert))
 *   purge         receiver/message_store.c:220-262 (drop messages older than 72 h)
 *
 * libsqlite3 is bound at run time with dlopen: the image ships the shared
 * library (python's sqlite3 uses it) but no sqlite3.h, so the dozen entry
 * points used here are declared from SQLite's published C interface.  A
 * missing library is an error (NVX_ERR_IO), never a silent no-op.
 *
 * Like the reference the database is opened and closed around every
 * operation, so the web server's own connections never meet a long-held lock.
 */
#define _GNU_SOURCE
#include "navtex_amd.h"
#include "nvx_internal.h"

#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct sqlite3 sqlite3;
typedef struct sqlite3_stmt sqlite3_stmt;
#define NVX_SQLITE_OK   0
#define NVX_SQLITE_ROW  100
#define NVX_SQLITE_DONE 101
typedef void (*sq_destructor)(void *);
#define NVX_SQLITE_TRANSIENT ((sq_destructor)-1)

static struct {
    void *dl;
    int  (*open)(const char *, sqlite3 **);
    int  (*close)(sqlite3 *);
    int  (*exec)(sqlite3 *, const char *, int (*)(void *, int, char **, char **), void *, char **);
    int  (*prepare_v2)(sqlite3 *, const char *, int, sqlite3_stmt **, const char **);
    int  (*bind_text)(sqlite3_stmt *, int, const char *, int, sq_destructor);
    int  (*bind_int)(sqlite3_stmt *, int, int);
    int  (*step)(sqlite3_stmt *);
    int  (*finalize)(sqlite3_stmt *);
    int  (*column_int)(sqlite3_stmt *, int);
    const unsigned char *(*column_text)(sqlite3_stmt *, int);
    const char *(*errmsg)(sqlite3 *);
    int  (*busy_timeout)(sqlite3 *, int);
    void (*free)(void *);
} sq;
static pthread_once_t sq_once = PTHREAD_ONCE_INIT;
static int sq_ok = 0;

static void sq_bind(void)
{
    const char *names[] = { getenv("NAVTEX_AMD_SQLITE"), "libsqlite3.so.0", "libsqlite3.so" };
    for (size_t i = 0; i < sizeof names / sizeof *names && !sq.dl; i++)
        if (names[i] && *names[i]) sq.dl = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!sq.dl) return;
#define SQ_SYM(field, name) do { *(void **)&sq.field = dlsym(sq.dl, name); if (!sq.field) return; } while (0)
    SQ_SYM(open, "sqlite3_open");             SQ_SYM(close, "sqlite3_close");
    SQ_SYM(exec, "sqlite3_exec");             SQ_SYM(prepare_v2, "sqlite3_prepare_v2");
    SQ_SYM(bind_text, "sqlite3_bind_text");   SQ_SYM(bind_int, "sqlite3_bind_int");
    SQ_SYM(step, "sqlite3_step");             SQ_SYM(finalize, "sqlite3_finalize");
    SQ_SYM(column_int, "sqlite3_column_int"); SQ_SYM(column_text, "sqlite3_column_text");
    SQ_SYM(errmsg, "sqlite3_errmsg");         SQ_SYM(busy_timeout, "sqlite3_busy_timeout");
    SQ_SYM(free, "sqlite3_free");
#undef SQ_SYM
    sq_ok = 1;
}

struct nvx_store {
    char *path;
    int64_t fixed_time;              /* test hook: != 0 replaces time(NULL) */
    pthread_mutex_t mu;              /* handles on different threads may share one store */
    uint64_t added, failed;
};

static time_t store_now(const nvx_store *s) { return s->fixed_time ? (time_t)s->fixed_time : time(NULL); }

static void format_ts(time_t t, char out[20])          /* message_store.c:26-34 */
{
    struct tm tm;
    gmtime_r(&t, &tm);
    strftime(out, 20, "%Y-%m-%d %H:%M", &tm);
}

static sqlite3 *store_open_db(const nvx_store *s)
{
    sqlite3 *db = NULL;
    if (sq.open(s->path, &db) != NVX_SQLITE_OK) {
        nvx_set_error("sqlite: cannot open %s: %s", s->path, db ? sq.errmsg(db) : "out of memory");
        if (db) sq.close(db);
        return NULL;
    }
    sq.busy_timeout(db, 2000);       /* the web server may be reading */
    return db;
}

static int store_exec(sqlite3 *db, const char *sql)
{
    char *err = NULL;
    if (sq.exec(db, sql, NULL, NULL, &err) != NVX_SQLITE_OK) {
        nvx_set_error("sqlite: %s", err ? err : sq.errmsg(db));
        if (err) sq.free(err);
        return NVX_ERR_IO;
    }
    return NVX_OK;
}

static int table_exists(sqlite3 *db, const char *name)
{
    sqlite3_stmt *st = NULL;
    int found = 0;
    if (sq.prepare_v2(db, "select 1 from sqlite_master where type='table' and name=?1", -1, &st, NULL) != NVX_SQLITE_OK) return 0;
    sq.bind_text(st, 1, name, -1, NVX_SQLITE_TRANSIENT);
    if (sq.step(st) == NVX_SQLITE_ROW) found = 1;
    sq.finalize(st);
    return found;
}

int nvx_store_open(const char *path, int create_schema, nvx_store **out)
{
    if (!path || !*path || !out) { nvx_set_error("nvx_store_open: bad argument"); return NVX_ERR_ARG; }
    *out = NULL;
    pthread_once(&sq_once, sq_bind);
    if (!sq_ok) {
        const char *why = dlerror();              /* dlerror() clears its message: read it once */
        nvx_set_error("nvx_store_open: libsqlite3.so.0 not found (%s)", why ? why : "missing symbol");
        return NVX_ERR_IO;
    }
    nvx_store *s = (nvx_store *)calloc(1, sizeof *s);
    if (!s) return NVX_ERR_NOMEM;
    s->path = strdup(path);
    if (!s->path) { free(s); return NVX_ERR_NOMEM; }
    pthread_mutex_init(&s->mu, NULL);
    sqlite3 *db = store_open_db(s);
    int rc = db ? NVX_OK : NVX_ERR_IO;
    if (db && create_schema) {                          /* generate_db.sql:3-8 */
        const int had_config = table_exists(db, "config");
        rc = store_exec(db, "CREATE TABLE IF NOT EXISTS \"messages\" (id integer primary key autoincrement,"
                            "bbbb text,message text,timestamp text, age text, freq integer);");
        if (rc == NVX_OK && !had_config)
            rc = store_exec(db, "BEGIN;"
                                "CREATE TABLE config (id integer primary key autoincrement,tag text,value text);"
                                "INSERT INTO config VALUES(1,'stations518','PSTV');"
                                "INSERT INTO config VALUES(2,'messages518','ABCDEFL');"
                                "INSERT INTO config VALUES(3,'stations490','B');"
                                "INSERT INTO config VALUES(4,'messages490','ABCDEFL');"
                                "COMMIT;");
    } else if (db && !table_exists(db, "messages")) {
        nvx_set_error("nvx_store_open: %s has no messages table (open with create_schema)", path);
        rc = NVX_ERR_IO;
    }
    if (db) sq.close(db);
    if (rc != NVX_OK) { nvx_store_close(s); return rc; }
    *out = s;
    return NVX_OK;
}

void nvx_store_close(nvx_store *s)
{
    if (!s) return;
    pthread_mutex_destroy(&s->mu);
    free(s->path);
    free(s);
}

void nvx_store_set_time(nvx_store *s, int64_t unix_seconds) { if (s) s->fixed_time = unix_seconds; }

void nvx_store_stats(nvx_store *s, uint64_t *added, uint64_t *failed)
{
    if (!s) return;
    pthread_mutex_lock(&s->mu);
    if (added) *added = s->added;
    if (failed) *failed = s->failed;
    pthread_mutex_unlock(&s->mu);
}

int nvx_store_add_message(nvx_store *s, const char *bbbb, const char *message, int freq)
{
    if (!s || !bbbb || !message) return -1;
    char ts[20];
    int rc = -2;
    pthread_mutex_lock(&s->mu);
    format_ts(store_now(s), ts);
    sqlite3 *db = store_open_db(s);
    if (!db) { s->failed++; pthread_mutex_unlock(&s->mu); return -1; }
    sqlite3_stmt *st = NULL;
    /* a repeat of the same broadcast replaces the earlier copy (message_store.c:73-76) */
    if (sq.prepare_v2(db, "delete from messages where bbbb = ?1 ", -1, &st, NULL) == NVX_SQLITE_OK) {
        sq.bind_text(st, 1, bbbb, -1, NVX_SQLITE_TRANSIENT);
        sq.step(st);
    }
    sq.finalize(st); st = NULL;
    if (sq.prepare_v2(db, "insert into messages (bbbb,message,timestamp,age,freq) values ( ?1 , ?2 , ?3 ,'NEW',?4)", -1, &st, NULL) == NVX_SQLITE_OK) {
        sq.bind_text(st, 1, bbbb, -1, NVX_SQLITE_TRANSIENT);
        sq.bind_text(st, 2, message, -1, NVX_SQLITE_TRANSIENT);
        sq.bind_text(st, 3, ts, -1, NVX_SQLITE_TRANSIENT);
        sq.bind_int(st, 4, freq);
        if (sq.step(st) == NVX_SQLITE_DONE) rc = 0;
        else nvx_set_error("sqlite: insert failed: %s", sq.errmsg(db));
    } else {
        nvx_set_error("sqlite: %s", sq.errmsg(db));
    }
    sq.finalize(st);
    sq.close(db);
    if (rc == 0) s->added++; else s->failed++;
    pthread_mutex_unlock(&s->mu);
    return rc;
}

void nvx_store_on_message(void *user, int stream, const char *bbbb, const char *message, int freq)
{
    (void)stream;
    nvx_store_add_message((nvx_store *)user, bbbb, message, freq);
}

/* "%Y-%m-%d %H:%M" (UTC) -> unix seconds; -1 if it does not parse */
static int64_t parse_ts(const char *ts)
{
    struct tm tm;
    memset(&tm, 0, sizeof tm);
    if (!ts || !strptime(ts, "%Y-%m-%d %H:%M", &tm)) return -1;
    return (int64_t)timegm(&tm);
}

int nvx_store_purge(nvx_store *s, long max_age_seconds)
{
    if (!s) return NVX_ERR_ARG;
    if (max_age_seconds <= 0) max_age_seconds = 60L * 60 * 72;          /* MESSAGE_PURGE_AGE, message_store.c:12 */
    pthread_mutex_lock(&s->mu);
    sqlite3 *db = store_open_db(s);
    if (!db) { pthread_mutex_unlock(&s->mu); return NVX_ERR_IO; }
    const int64_t now = (int64_t)store_now(s);
    int purged = 0, *ids = NULL, n = 0, cap = 0;
    sqlite3_stmt *st = NULL;
    if (sq.prepare_v2(db, "select id,timestamp from messages order by timestamp asc", -1, &st, NULL) != NVX_SQLITE_OK) {
        nvx_set_error("sqlite: %s", sq.errmsg(db));
        sq.close(db); pthread_mutex_unlock(&s->mu);
        return NVX_ERR_IO;
    }
    while (sq.step(st) == NVX_SQLITE_ROW) {
        int64_t t = parse_ts((const char *)sq.column_text(st, 1));
        if (t < 0 || now - t <= (int64_t)max_age_seconds) continue;
        if (n == cap) {
            int *grown = (int *)realloc(ids, (size_t)(cap = cap ? 2 * cap : 64) * sizeof *ids);
            if (!grown) break;
            ids = grown;
        }
        ids[n++] = sq.column_int(st, 0);
    }
    sq.finalize(st);
    for (int i = 0; i < n; i++) {
        st = NULL;
        if (sq.prepare_v2(db, "delete from messages where id = ?1 ", -1, &st, NULL) == NVX_SQLITE_OK) {
            sq.bind_int(st, 1, ids[i]);
            if (sq.step(st) == NVX_SQLITE_DONE) purged++;
        }
        sq.finalize(st);
    }
    free(ids);
    sq.close(db);
    pthread_mutex_unlock(&s->mu);
    return purged;
}
