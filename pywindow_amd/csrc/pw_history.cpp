// pw_history.cpp -- native DL_POLY HISTORY ingest (host side of the hot path).
//
// Counterpart of DLPOLY._map_history / _decode_head / _decode_frame
// (reference trajectory.py:647-766): the file is mmap'ed once, every line that
// starts with the token "timestep" opens a frame, and a frame is
//   timestep nstep natms keytrj imcon tstep
//   [3 lattice lines when imcon in 1..3]
//   per atom: a key line, a coordinate line, (+ velocity line if keytrj >= 1,
//   + force line if keytrj == 2)
// Numbers are converted with strtod (correctly rounded, the same values
// Python's float() yields in the reference).  The Python parser costs
// ~1.3 ms/frame; this one runs at memory speed and fills
// contiguous (frames, atoms, 3) buffers that go straight to the GPU.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <pthread.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pywindow_amd.h"

// (pw_kernels.hip) the calling thread's error text, as pw_last_error() returns it
extern "C" void pw_internal_set_error(const char* msg);

struct pw_history {
    int fd;
    const char* data;
    size_t size;
    int keytrj, imcon;
    int64_t natoms;
    std::vector<size_t> frame_start;  // byte offset of each "timestep" line
    std::vector<size_t> frame_end;
};

namespace {

// ---- reader threads -------------------------------------------------------------------------------------
// Indexing a file and decoding a block of frames are cut into independent ranges for up to 16 host threads.
// The threads are started ONCE per process and parked on a condition variable between calls (spawning and
// joining sixteen std::threads cost 0.2-0.4 ms of every call -- a tenth of a 1000-frame analysis end to end);
// the calling thread takes range 0 itself.  One parallel region at a time: a second caller that finds the
// team busy runs its ranges on threads of its own, as every call used to.  A forked child starts a fresh team.
class ReaderTeam {
  public:
    static constexpr int MAXT = 16;
    static int max_threads() {
        unsigned hw = std::thread::hardware_concurrency();
        int n = hw ? (int)hw : 1;
        if (const char* e = getenv("PW_READER_THREADS")) if (atoi(e) > 0) n = atoi(e);
        return n > MAXT ? MAXT : n;
    }
    // fn(t) for t in [0, n): returns when all have run.  Range t belongs to thread t (the caller is thread 0) -- so a
    // region is spread over the threads however unevenly they wake up: a worker that came back early used to take
    // every remaining range while the others were still waking, and a read ran serially -- and whoever finishes its
    // own range takes over ranges nobody has started yet, so one slow thread does not hold the region up either.
    // An exception thrown by fn is rethrown here once every range has ended.
    static void run(int n, const std::function<void(int)>& fn) {
        if (n <= 1) { if (n == 1) fn(0); return; }
        if (n > MAXT) {
            // more ranges than the team has threads (no caller asks for that today: max_threads() is what they cut
            // their work by): the first MAXT - 1 side by side with the rest of them, in turn, on the calling thread's
            // range -- never a range that silently does not run
            const int extra_first = MAXT - 1;
            const std::function<void(int)> folded = [&](int t) {
                if (t < extra_first) { fn(t); return; }
                for (int k = extra_first; k < n; ++k) fn(k);
            };
            run(MAXT, folded);
            return;
        }
        ReaderTeam* team = instance();
        std::unique_lock<std::mutex> region(team->region_, std::try_to_lock);
        if (!region.owns_lock() || !team->start(n - 1)) {
            std::vector<std::thread> own;
            for (int t = 1; t < n; ++t) own.emplace_back(fn, t);
            fn(0);
            for (auto& th : own) th.join();
            return;
        }
        {
            std::lock_guard<std::mutex> g(team->m_);
            team->fn_ = &fn;
            team->n_ = n;
            team->left_ = n;
            team->error_ = nullptr;
            team->epoch_ += 1;
        }
        team->wake_.notify_all();
        team->work(0);
        std::exception_ptr err;
        {
            std::unique_lock<std::mutex> g(team->m_);
            team->done_.wait(g, [&] { return team->left_ == 0; });
            team->fn_ = nullptr;
            team->n_ = 0;
            err = team->error_;
            team->error_ = nullptr;
        }
        if (err) std::rethrow_exception(err);
    }

  private:
    static ReaderTeam*& slot() { static ReaderTeam* t = nullptr; return t; }
    static ReaderTeam* instance() {
        static std::once_flag once;
        std::call_once(once, [] {
            slot() = new ReaderTeam();
            // the threads do not exist in a forked child: it gets a team of its own on first use
            pthread_atfork(nullptr, nullptr, [] { slot() = new ReaderTeam(); });
        });
        return slot();
    }
    bool start(int want) {          // (caller holds region_)
        try {
            while ((int)threads_.size() < want && (int)threads_.size() < MAXT - 1) {
                const int id = (int)threads_.size() + 1;
                threads_.emplace_back([this, id] { loop(id); });
                threads_.back().detach();
            }
        } catch (...) {
        }
        return (int)threads_.size() >= want;
    }
    // thread `me` of the current region: its own range first, then whatever nobody has claimed
    void work(int me) {
        const std::function<void(int)>* fn;
        int n;
        unsigned long e;
        {
            std::lock_guard<std::mutex> g(m_);
            fn = fn_;
            n = n_;
            e = epoch_;
        }
        if (!fn) return;
        int finished = 0;
        std::exception_ptr err;
        for (int k = 0; k < n; ++k) {
            const int t = (me + k) % n;
            if (me >= n && k == 0) continue;             // (a worker beyond this region's width only helps out)
            // a range is claimed by raising its mark to the region's number: a thread that is still holding an older
            // region's function (it woke up late) can never claim a range of this one
            unsigned long cur = claimed_[t].load(std::memory_order_acquire);
            bool mine = false;
            while (cur < e && !(mine = claimed_[t].compare_exchange_weak(cur, e, std::memory_order_acq_rel))) {}
            if (!mine) continue;
            try {
                (*fn)(t);
            } catch (...) {
                if (!err) err = std::current_exception();
            }
            finished += 1;
        }
        if (finished || err) {
            std::lock_guard<std::mutex> g(m_);
            if (err && !error_) error_ = err;
            left_ -= finished;
            if (left_ == 0) done_.notify_all();
        }
    }
    void loop(int me) {
        unsigned long seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                wake_.wait(g, [&] { return epoch_ != seen; });
                seen = epoch_;
            }
            work(me);
        }
    }
    std::mutex region_, m_;
    std::condition_variable wake_, done_;
    std::vector<std::thread> threads_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<unsigned long> claimed_[MAXT] = {};
    std::exception_ptr error_;
    int n_ = 0, left_ = 0;
    unsigned long epoch_ = 0;
};

inline const char* line_end(const char* p, const char* end) {
    const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
    return q ? q : end;
}
inline const char* skip_ws(const char* p, const char* e) {
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    return p;
}
inline bool first_token_is(const char* p, const char* e, const char* tok) {
    p = skip_ws(p, e);
    size_t n = strlen(tok);
    if ((size_t)(e - p) < n || memcmp(p, tok, n) != 0) return false;
    const char* q = p + n;
    return q == e || *q == ' ' || *q == '\t' || *q == '\r';
}
// One decimal token -> double.  Fast path (Clinger): a mantissa below 2^53 and a power of ten up
// to 10^22 are both exact doubles, so one multiplication or division is correctly rounded --
// the value strtod / Python's float() give.  Anything else goes to strtod.
inline bool parse_one(const char* p, const char* q, double* out) {
    static const double P10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                   1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* c = p;
    bool neg = false;
    if (c < q && (*c == '+' || *c == '-')) { neg = *c == '-'; ++c; }
    uint64_t m = 0;
    int digits = 0, frac = 0;
    bool seen_dot = false, any = false, fast = true;
    for (; c < q; ++c) {
        if (*c >= '0' && *c <= '9') {
            any = true;
            if (digits < 18) { m = m * 10 + (uint64_t)(*c - '0'); if (m) ++digits; if (seen_dot) ++frac; }
            else fast = false;
        } else if (*c == '.' && !seen_dot) {
            seen_dot = true;
        } else {
            break;
        }
    }
    int e10 = 0;
    if (any && c < q && (*c == 'e' || *c == 'E' || *c == 'd' || *c == 'D')) {
        if (*c == 'd' || *c == 'D') fast = false;        // Fortran exponent: let strtod reject it like float()
        ++c;
        bool eneg = false;
        if (c < q && (*c == '+' || *c == '-')) { eneg = *c == '-'; ++c; }
        int ev = 0, nd = 0;
        for (; c < q && *c >= '0' && *c <= '9'; ++c, ++nd) if (ev < 10000) ev = ev * 10 + (*c - '0');
        if (nd == 0) fast = false;
        e10 = eneg ? -ev : ev;
    }
    if (any && fast && c == q && m < (1ull << 53)) {
        int e = e10 - frac;
        if (e >= -22 && e <= 22) {
            double v = (double)m;
            v = e >= 0 ? v * P10[e] : v / P10[-e];
            *out = neg ? -v : v;
            return true;
        }
    }
    char buf[64];
    size_t n = (size_t)(q - p);
    if (n >= sizeof(buf)) return false;
    memcpy(buf, p, n);
    buf[n] = 0;
    char* endp = nullptr;
    double v = strtod(buf, &endp);
    if (endp == buf || *endp != 0) return false;
    *out = v;
    return true;
}
// parse up to `want` whitespace separated doubles from [p, e)
inline int parse_doubles(const char* p, const char* e, double* out, int want) {
    int got = 0;
    while (got < want) {
        p = skip_ws(p, e);
        if (p >= e) break;
        const char* q = p;
        while (q < e && *q != ' ' && *q != '\t' && *q != '\r') ++q;
        if (!parse_one(p, q, &out[got])) return -1;
        ++got;
        p = q;
    }
    return got;
}

// iterate the lines of one frame; calls back with (kind, begin, end): kind 0 = key line,
// 1 = coordinate line, 2 = lattice line
template <class F>
static int walk_frame(const pw_history* h, int64_t f, F&& cb) {
    const char* p = h->data + h->frame_start[(size_t)f];
    const char* end = h->data + h->frame_end[(size_t)f];
    const char* le = line_end(p, end);
    // timestep nstep natms keytrj imcon tstep
    const char* q = skip_ws(p, le);
    while (q < le && *q != ' ' && *q != '\t') ++q;  // skip the word
    double v[5];
    if (parse_doubles(q, le, v, 5) != 5) return PW_E_BAD_ARG;
    int64_t natms = (int64_t)v[1];
    int keytrj = (int)v[2], imcon = (int)v[3];
    p = le < end ? le + 1 : end;
    if (imcon >= 1 && imcon <= 3) {
        for (int r = 0; r < 3; ++r) {
            le = line_end(p, end);
            if (cb(2, r, p, le) != 0) return PW_E_BAD_ARG;
            p = le < end ? le + 1 : end;
        }
    }
    int per_atom = 2 + keytrj;
    for (int64_t a = 0; a < natms; ++a) {
        for (int r = 0; r < per_atom; ++r) {
            if (p >= end) return PW_E_BAD_ARG;
            le = line_end(p, end);
            if (r < 2 && cb(r, (int)a, p, le) != 0) return PW_E_BAD_ARG;
            p = le < end ? le + 1 : end;
        }
    }
    return (int)natms == natms ? PW_OK : PW_E_BAD_ARG;
}

}  // namespace

extern "C" {

int pw_history_open(const char* path, pw_history** out) {
    if (!path || !out) return PW_E_BAD_ARG;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return PW_E_BAD_ARG;
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size == 0) {
        close(fd);
        return PW_E_BAD_ARG;
    }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);   // page tables in one go
    if (m == MAP_FAILED) {
        close(fd);
        return PW_E_NOMEM;
    }
    pw_history* h = new (std::nothrow) pw_history();
    if (!h) {
        munmap(m, (size_t)st.st_size);
        close(fd);
        return PW_E_NOMEM;
    }
    h->fd = fd;
    h->data = (const char*)m;
    h->size = (size_t)st.st_size;
    h->keytrj = 0;
    h->imcon = 0;
    h->natoms = 0;
    const char* end = h->data + h->size;
    {
        // header: the second line carries keytrj, imcon, natms
        const char* l1 = line_end(h->data, end);
        if (l1 < end) {
            const char* p2 = l1 + 1;
            const char* l2 = line_end(p2, end);
            double v[3];
            if (!first_token_is(p2, l2, "timestep") && parse_doubles(p2, l2, v, 3) == 3) {
                h->keytrj = (int)v[0];
                h->imcon = (int)v[1];
                h->natoms = (int64_t)v[2];
            }
        }
    }
    // frame starts: lines whose first token is "timestep".  The file is cut into byte ranges, one
    // host thread each; a range begins at the first line start at or after its first byte.
    int nthreads = ReaderTeam::max_threads();
    if ((size_t)nthreads > h->size / (1u << 20)) nthreads = (int)(h->size / (1u << 20));
    if (nthreads < 1) nthreads = 1;
    std::vector<std::vector<size_t>> found((size_t)nthreads);
    auto scan = [&](int t) {
        const char* lo = h->data + h->size * (size_t)t / (size_t)nthreads;
        const char* hi = h->data + h->size * (size_t)(t + 1) / (size_t)nthreads;
        const char* p = lo;
        if (t > 0 && p[-1] != '\n') {                      // not at a line start: skip the partial line
            p = line_end(p, end);
            p = p < end ? p + 1 : end;
        }
        while (p < hi) {
            const char* le = line_end(p, end);
            if (first_token_is(p, le, "timestep")) found[(size_t)t].push_back((size_t)(p - h->data));
            p = le < end ? le + 1 : end;
        }
    };
    ReaderTeam::run(nthreads, scan);
    for (auto& part : found)
        for (size_t off : part) {
            if (!h->frame_start.empty()) h->frame_end.push_back(off);
            h->frame_start.push_back(off);
        }
    if (!h->frame_start.empty()) h->frame_end.push_back(h->size);
    *out = h;
    return PW_OK;
}

int64_t pw_history_frames(const pw_history* h) { return h ? (int64_t)h->frame_start.size() : 0; }
int64_t pw_history_atoms(const pw_history* h) { return h ? h->natoms : 0; }
int pw_history_keytrj(const pw_history* h) { return h ? h->keytrj : 0; }
int pw_history_imcon(const pw_history* h) { return h ? h->imcon : 0; }

int64_t pw_history_atom_keys(const pw_history* h, char* buf, int64_t buflen) {
    if (!h || h->frame_start.empty()) return 0;
    std::string keys;
    int rc = walk_frame(h, 0, [&](int kind, int, const char* b, const char* e) {
        if (kind == 0) {
            b = skip_ws(b, e);
            const char* q = b;
            while (q < e && *q != ' ' && *q != '\t' && *q != '\r') ++q;
            keys.append(b, (size_t)(q - b));
            keys.push_back('\0');
        }
        return 0;
    });
    if (rc != PW_OK) return rc;
    if (buf && buflen >= (int64_t)keys.size()) memcpy(buf, keys.data(), keys.size());
    return (int64_t)keys.size();
}

static int read_range(const pw_history* h, int64_t first, int64_t f0, int64_t f1, double* xyz, double* lattice) {
    const int64_t n = h->natoms;
    for (int64_t f = f0; f < f1; ++f) {
        double* dst = xyz + (size_t)f * (size_t)n * 3;
        double* lat = lattice ? lattice + (size_t)f * 9 : nullptr;
        int64_t seen = 0;
        int rc = walk_frame(h, first + f, [&](int kind, int idx, const char* b, const char* e) {
            if (kind == 1) {
                if (idx >= n) return -1;
                if (parse_doubles(b, e, dst + 3 * (size_t)idx, 3) != 3) return -1;
                ++seen;
            } else if (kind == 2 && lat) {
                // rows of the file are the lattice vectors; the reference stores the transpose
                double v[3];
                if (parse_doubles(b, e, v, 3) != 3) return -1;
                lat[0 * 3 + idx] = v[0];
                lat[1 * 3 + idx] = v[1];
                lat[2 * 3 + idx] = v[2];
            }
            return 0;
        });
        if (rc != PW_OK || seen != n) return PW_E_BAD_ARG;
    }
    return PW_OK;
}

int pw_history_read(const pw_history* h, int64_t first, int64_t count, double* xyz, double* lattice) {
    if (!h || !xyz || first < 0 || count < 0 || first + count > (int64_t)h->frame_start.size())
        return PW_E_BAD_ARG;
    // frames are independent: decode them on several host threads
    int64_t nthreads = ReaderTeam::max_threads();
    if (nthreads > count / 16) nthreads = count / 16;
    if (nthreads <= 1) return read_range(h, first, 0, count, xyz, lattice);
    std::vector<int> rcs((size_t)nthreads, PW_OK);
    ReaderTeam::run((int)nthreads, [&](int t) {
        const int64_t f0 = count * t / nthreads, f1 = count * (t + 1) / nthreads;
        rcs[(size_t)t] = read_range(h, first, f0, f1, xyz, lattice);
    });
    for (int rc : rcs) if (rc != PW_OK) return rc;
    return PW_OK;
}

int pw_history_reader_threads(void) { return ReaderTeam::max_threads(); }

// Frames [first_frame, first_frame + count) decoded straight into `staging` (the context's page-locked buffer, count x
// atoms x 3) and handed to the streamed batch `res` WHILE the decoding goes on: the reader's threads take blocks of
// eight frames from a counter, and one more thread watches the prefix of finished blocks and appends it
// (pw_resident_stream_append: a DMA copy, then the launch's `ready` counter) whenever at least `min_append` more frames
// are complete -- the analysis of the first frames runs while the last are still text, the copies overlap the decoding,
// and what is left to do once the reader has finished is the analysis of the last few dozen frames, not of a quarter
// of the trajectory.  legs_ms (may be null): [0] until every frame was decoded, [1] from there until the last append
// had returned.  The counterpart of the frame loop of Trajectory._analysis_serial (trajectory.py:496-522).
int pw_history_stream_read(const pw_history* h, int64_t first_frame, int64_t count, pw_context* ctx, pw_resident* res,
                           int64_t first_unit, double* staging, int64_t min_append, double* legs_ms) {
    if (!h || !ctx || !res || !staging || first_frame < 0 || count < 0 || first_unit < 0 ||
        first_frame + count > (int64_t)h->frame_start.size())
        return PW_E_BAD_ARG;
    if (count == 0) return PW_OK;
    if (min_append < 1) min_append = 64;
    constexpr int64_t B = 8;
    const int64_t nblocks = (count + B - 1) / B;
    const size_t per = (size_t)h->natoms * 3;
    std::vector<std::atomic<int>> done((size_t)nblocks);
    for (auto& d : done) d.store(0, std::memory_order_relaxed);
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{PW_OK};
    std::atomic<int> decoded_all{0};
    std::atomic<int64_t> bad_frame{-1};
    const auto t0 = std::chrono::steady_clock::now();
    int append_rc = PW_OK;
    std::string append_err;              // the appender thread's error text (pw_last_error is per thread)
    // the appender sleeps until a decoder has finished a block (or 200 us have passed): it used to spin on yield(),
    // a core of the reader's own on machines with no more cores than reader threads
    std::mutex bm;
    std::condition_variable bcv;
    auto appends = [&] {
        int64_t appended = 0, p = 0;
        for (;;) {
            const bool all = decoded_all.load(std::memory_order_acquire) != 0;
            while (p < nblocks && done[(size_t)p].load(std::memory_order_acquire)) ++p;
            const int64_t have = p * B < count ? p * B : count;
            if (have - appended >= min_append || (have == count && have > appended)) {
                append_rc = pw_resident_stream_append(ctx, res, staging + (size_t)appended * per, first_unit + appended, have - appended);
                if (append_rc != PW_OK) { append_err = pw_last_error(); return; }
                appended = have;
                if (appended == count) return;
                continue;
            }
            if (failed.load(std::memory_order_acquire) != PW_OK || (all && p < nblocks)) return;   // (a block that never finished)
            std::unique_lock<std::mutex> g(bm);
            bcv.wait_for(g, std::chrono::microseconds(200));
        }
    };
    std::thread appender;
    bool have_appender = true;
    try {
        appender = std::thread(appends);
    } catch (...) {
        have_appender = false;         // (no thread to be had: decode everything, then append)
    }
    int64_t nthreads = ReaderTeam::max_threads();
    if (nthreads > nblocks) nthreads = nblocks;
    ReaderTeam::run((int)nthreads, [&](int) {
        for (;;) {
            const int64_t b = next.fetch_add(1, std::memory_order_relaxed);
            if (b >= nblocks || failed.load(std::memory_order_relaxed) != PW_OK) return;
            const int64_t f0 = b * B, f1 = f0 + B < count ? f0 + B : count;
            const int rc = read_range(h, first_frame, f0, f1, staging, nullptr);
            if (rc != PW_OK) {
                int64_t none = -1;
                bad_frame.compare_exchange_strong(none, first_frame + f0);
                failed.store(rc, std::memory_order_release);
                bcv.notify_all();
                return;
            }
            done[(size_t)b].store(1, std::memory_order_release);
            bcv.notify_all();
        }
    });
    decoded_all.store(1, std::memory_order_release);
    bcv.notify_all();
    const auto t1 = std::chrono::steady_clock::now();
    if (have_appender) appender.join();
    else appends();
    const auto t2 = std::chrono::steady_clock::now();
    if (legs_ms) {
        legs_ms[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
        legs_ms[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    }
    // the failing thread's reason, in the CALLING thread's error text (the appends ran on a thread of their own, the
    // decoders on the reader's threads)
    const int frc = failed.load(std::memory_order_acquire);
    if (frc != PW_OK) {
        char msg[160];
        snprintf(msg, sizeof(msg), "HISTORY: a frame in the block starting at frame %lld cannot be decoded (%lld atoms expected)",
                 (long long)bad_frame.load(), (long long)h->natoms);
        pw_internal_set_error(msg);
        return frc;
    }
    if (append_rc != PW_OK) pw_internal_set_error(append_err.c_str());
    return append_rc;
}

// nstep and tstep of the "timestep" record of frame f (the reference keeps them as frame_info,
// trajectory.py:712-721)
int pw_history_frame_info(const pw_history* h, int64_t frame, int64_t* nstep, double* tstep) {
    if (!h || frame < 0 || frame >= (int64_t)h->frame_start.size()) return PW_E_BAD_ARG;
    const char* p = h->data + h->frame_start[(size_t)frame];
    const char* end = h->data + h->frame_end[(size_t)frame];
    const char* le = line_end(p, end);
    const char* q = skip_ws(p, le);
    while (q < le && *q != ' ' && *q != '\t') ++q;
    double v[5];
    if (parse_doubles(q, le, v, 5) != 5) return PW_E_BAD_ARG;
    if (nstep) *nstep = (int64_t)v[0];
    if (tstep) *tstep = v[4];
    return PW_OK;
}

void pw_history_close(pw_history* h) {
    if (!h) return;
    if (h->data) munmap((void*)h->data, h->size);
    if (h->fd >= 0) close(h->fd);
    delete h;
}

}  // extern "C"
