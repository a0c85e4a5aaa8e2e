// io.cpp -- rating-file readers and factor files (the data formats
// either side of the hot path).  Host only.  No reference counterpart exists
// (/root/reference/README.md:1-2); the formats are the public ones of the datasets
// BASELINE.json's configs are shaped after:
//   MFSGD_FMT_ML_TSV   MovieLens-100K  u.data        "user \t item \t rating \t timestamp"
//   MFSGD_FMT_ML_DAT   MovieLens-1M/10M ratings.dat  "user::movie::rating::timestamp"
//   MFSGD_FMT_ML_CSV   MovieLens-20M/25M ratings.csv "userId,movieId,rating,timestamp" (+ header)
//   MFSGD_FMT_NETFLIX  Netflix Prize combined_data_N.txt / mv_*.txt: "movieId:" lines followed
//                      by "customerId,rating,date" lines
// Original ids are arbitrary (and sparse for movies), so they are compacted: dense index =
// rank of the id among the distinct ids of the file (ascending), and the id tables are kept.
#include <algorithm>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mfsgd.h"

struct mfsgd_ratings_file {
    std::vector<int32_t> u, i;
    std::vector<float> r;
    std::vector<int64_t> user_ids, item_ids;  // dense index -> original id
};

namespace {

thread_local std::string g_io_error;

int io_fail(int code, const std::string& msg) {
    g_io_error = msg;
    return code;
}

bool read_whole(const char* path, std::vector<char>& buf, std::string& err) {
    FILE* f = std::fopen(path, "rb");
    if (!f) {
        err = std::string("cannot open ") + path + ": " + std::strerror(errno);
        return false;
    }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) {
        std::fclose(f);
        err = "ftell failed";
        return false;
    }
    buf.resize((size_t)n + 1);
    const size_t got = n ? std::fread(buf.data(), 1, (size_t)n, f) : 0;
    std::fclose(f);
    if (got != (size_t)n) {
        err = "short read";
        return false;
    }
    buf[(size_t)n] = '\0';
    return true;
}

int detect_format(const char* p) {
    // first non-empty line decides
    while (*p == '\n' || *p == '\r') ++p;
    const char* e = p;
    while (*e && *e != '\n') ++e;
    const std::string line(p, e);
    if (line.find("::") != std::string::npos) return MFSGD_FMT_ML_DAT;
    if (line.find('\t') != std::string::npos) return MFSGD_FMT_ML_TSV;
    if (!line.empty() && line.find(',') == std::string::npos && line.find(':') != std::string::npos)
        return MFSGD_FMT_NETFLIX;
    if (line.find(',') != std::string::npos) return MFSGD_FMT_ML_CSV;
    return 0;
}

// parses "<int><sep><int><sep><float>[<sep>...]" lines; sep is one char, or "::" when sep == ':'
bool parse_triples(char* p, char sep, bool allow_header, std::vector<int64_t>& uid, std::vector<int64_t>& iid,
                   std::vector<float>& r, std::string& err) {
    int64_t lineno = 0;
    while (*p) {
        ++lineno;
        char* eol = p;
        while (*eol && *eol != '\n') ++eol;
        char* next = *eol ? eol + 1 : eol;
        if (eol > p && eol[-1] == '\r') --eol;
        if (eol == p) {
            p = next;
            continue;
        }
        char* q = p;
        char* end = nullptr;
        errno = 0;
        // strtoll skips leading white space, newlines included: a field that is empty up to the end
        // of its line must not be filled from the next one
        auto at_eol = [&](const char* x) {
            while (x < eol && (*x == ' ' || *x == '\t') && *x != sep) ++x;
            return x >= eol;
        };
        const long long a = at_eol(q) ? 0 : std::strtoll(q, &end, 10);
        if (at_eol(q) || end == q || end > eol) {
            if (allow_header && lineno == 1) {  // "userId,movieId,rating,timestamp"
                p = next;
                continue;
            }
            err = "line " + std::to_string(lineno) + ": expected a user id";
            return false;
        }
        q = end;
        auto skip_sep = [&]() {
            if (sep == ':') {
                if (q[0] == ':' && q[1] == ':') {
                    q += 2;
                    return true;
                }
                return false;
            }
            if (*q == sep) {
                ++q;
                return true;
            }
            return false;
        };
        if (!skip_sep()) {
            err = "line " + std::to_string(lineno) + ": missing separator after the user id";
            return false;
        }
        const long long b = at_eol(q) ? 0 : std::strtoll(q, &end, 10);
        if (at_eol(q) || end == q || end > eol) {
            err = "line " + std::to_string(lineno) + ": expected an item id";
            return false;
        }
        q = end;
        if (!skip_sep()) {
            err = "line " + std::to_string(lineno) + ": missing separator after the item id";
            return false;
        }
        const float x = std::strtof(q, &end);
        if (end == q || end > eol) {
            err = "line " + std::to_string(lineno) + ": expected a rating";
            return false;
        }
        uid.push_back(a);
        iid.push_back(b);
        r.push_back(x);
        p = next;
    }
    return true;
}

bool parse_netflix(char* p, std::vector<int64_t>& uid, std::vector<int64_t>& iid, std::vector<float>& r,
                   std::string& err) {
    int64_t lineno = 0, movie = -1;
    while (*p) {
        ++lineno;
        char* eol = p;
        while (*eol && *eol != '\n') ++eol;
        char* next = *eol ? eol + 1 : eol;
        if (eol > p && eol[-1] == '\r') --eol;
        if (eol == p) {
            p = next;
            continue;
        }
        char* end = nullptr;
        const long long a = std::strtoll(p, &end, 10);
        if (end == p) {
            err = "line " + std::to_string(lineno) + ": expected a number";
            return false;
        }
        if (*end == ':') {
            movie = a;
        } else if (*end == ',') {
            if (movie < 0) {
                err = "line " + std::to_string(lineno) + ": rating before any 'movieId:' line";
                return false;
            }
            char* q = end + 1;
            const float x = std::strtof(q, &end);
            if (end == q) {
                err = "line " + std::to_string(lineno) + ": expected a rating";
                return false;
            }
            uid.push_back(a);
            iid.push_back(movie);
            r.push_back(x);
        } else {
            err = "line " + std::to_string(lineno) + ": neither 'movieId:' nor 'customerId,rating,date'";
            return false;
        }
        p = next;
    }
    return true;
}

void compact(const std::vector<int64_t>& ids, std::vector<int64_t>& table, std::vector<int32_t>& dense) {
    table = ids;
    std::sort(table.begin(), table.end());
    table.erase(std::unique(table.begin(), table.end()), table.end());
    dense.resize(ids.size());
    for (size_t j = 0; j < ids.size(); ++j)
        dense[j] = (int32_t)(std::lower_bound(table.begin(), table.end(), ids[j]) - table.begin());
}

}  // namespace

extern "C" {

const char* mfsgd_io_last_error(void) { return g_io_error.c_str(); }

int mfsgd_ratings_file_open(const char* path, int32_t format, mfsgd_ratings_file** out) {
    if (out) *out = nullptr;
    if (!path || !out) return io_fail(MFSGD_ERR_INVALID_ARG, "ratings_file_open: null argument");
    try {
        std::vector<char> buf;
        std::string err;
        if (!read_whole(path, buf, err)) return io_fail(MFSGD_ERR_INVALID_ARG, err);
        if (format == MFSGD_FMT_AUTO) format = detect_format(buf.data());
        std::vector<int64_t> uid, iid;
        std::vector<float> r;
        bool ok = false;
        switch (format) {
            case MFSGD_FMT_ML_TSV: ok = parse_triples(buf.data(), '\t', false, uid, iid, r, err); break;
            case MFSGD_FMT_ML_DAT: ok = parse_triples(buf.data(), ':', false, uid, iid, r, err); break;
            case MFSGD_FMT_ML_CSV: ok = parse_triples(buf.data(), ',', true, uid, iid, r, err); break;
            case MFSGD_FMT_NETFLIX: ok = parse_netflix(buf.data(), uid, iid, r, err); break;
            default: return io_fail(MFSGD_ERR_INVALID_ARG, "ratings_file_open: unknown or undetectable format");
        }
        if (!ok) return io_fail(MFSGD_ERR_INVALID_ARG, std::string(path) + ": " + err);
        if (uid.size() > 0x7FFFFFF0ull * 64) return io_fail(MFSGD_ERR_UNSUPPORTED, "too many ratings");
        mfsgd_ratings_file* f = new mfsgd_ratings_file();
        compact(uid, f->user_ids, f->u);
        compact(iid, f->item_ids, f->i);
        f->r.swap(r);
        *out = f;
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return io_fail(MFSGD_ERR_OOM, "ratings_file_open: out of host memory");
    }
}

int mfsgd_ratings_file_info(const mfsgd_ratings_file* f, int64_t* nnz, int32_t* n_users, int32_t* n_items) {
    if (!f) return io_fail(MFSGD_ERR_INVALID_ARG, "ratings_file_info: null file");
    if (nnz) *nnz = (int64_t)f->r.size();
    if (n_users) *n_users = (int32_t)f->user_ids.size();
    if (n_items) *n_items = (int32_t)f->item_ids.size();
    return MFSGD_OK;
}

int mfsgd_ratings_file_read(const mfsgd_ratings_file* f, int32_t* u, int32_t* i, float* r, int64_t* user_ids,
                            int64_t* item_ids) {
    if (!f) return io_fail(MFSGD_ERR_INVALID_ARG, "ratings_file_read: null file");
    const size_t n = f->r.size();
    if (u && n) std::memcpy(u, f->u.data(), n * sizeof(int32_t));
    if (i && n) std::memcpy(i, f->i.data(), n * sizeof(int32_t));
    if (r && n) std::memcpy(r, f->r.data(), n * sizeof(float));
    if (user_ids && !f->user_ids.empty()) std::memcpy(user_ids, f->user_ids.data(), f->user_ids.size() * sizeof(int64_t));
    if (item_ids && !f->item_ids.empty()) std::memcpy(item_ids, f->item_ids.data(), f->item_ids.size() * sizeof(int64_t));
    return MFSGD_OK;
}

void mfsgd_ratings_file_close(mfsgd_ratings_file* f) { delete f; }

// ---- factor files: "MFSGDF01", int32 U, I, k, reserved, then P (U x k) and Q (I x k), fp32 LE ----
int mfsgd_save_factors(mfsgd_handle* h, const char* path) {
    if (!h || !path) return io_fail(MFSGD_ERR_INVALID_ARG, "save_factors: null argument");
    int32_t dims[3];
    int rc = mfsgd_get_dims(h, &dims[0], &dims[1], &dims[2]);
    if (rc) return rc;
    try {
        std::vector<float> P((size_t)dims[0] * dims[2]), Q((size_t)dims[1] * dims[2]);
        rc = mfsgd_get_factors(h, P.data(), Q.data());
        if (rc) return rc;
        FILE* f = std::fopen(path, "wb");
        if (!f) return io_fail(MFSGD_ERR_INVALID_ARG, std::string("cannot create ") + path + ": " + std::strerror(errno));
        const int32_t hdr[4] = {dims[0], dims[1], dims[2], 0};
        bool ok = std::fwrite("MFSGDF01", 1, 8, f) == 8 && std::fwrite(hdr, sizeof hdr, 1, f) == 1 &&
                  std::fwrite(P.data(), sizeof(float), P.size(), f) == P.size() &&
                  std::fwrite(Q.data(), sizeof(float), Q.size(), f) == Q.size();
        ok = (std::fclose(f) == 0) && ok;
        if (!ok) return io_fail(MFSGD_ERR_INVALID_ARG, std::string("write failed: ") + path);
        return MFSGD_OK;
    } catch (const std::bad_alloc&) {
        return io_fail(MFSGD_ERR_OOM, "save_factors: out of host memory");
    }
}

int mfsgd_load_factors(mfsgd_handle* h, const char* path) {
    if (!h || !path) return io_fail(MFSGD_ERR_INVALID_ARG, "load_factors: null argument");
    int32_t dims[3];
    int rc = mfsgd_get_dims(h, &dims[0], &dims[1], &dims[2]);
    if (rc) return rc;
    FILE* f = std::fopen(path, "rb");
    if (!f) return io_fail(MFSGD_ERR_INVALID_ARG, std::string("cannot open ") + path + ": " + std::strerror(errno));
    char magic[8];
    int32_t hdr[4];
    if (std::fread(magic, 1, 8, f) != 8 || std::memcmp(magic, "MFSGDF01", 8) != 0 || std::fread(hdr, sizeof hdr, 1, f) != 1) {
        std::fclose(f);
        return io_fail(MFSGD_ERR_INVALID_ARG, std::string(path) + ": not a factor file");
    }
    if (hdr[0] != dims[0] || hdr[1] != dims[1] || hdr[2] != dims[2]) {
        std::fclose(f);
        return io_fail(MFSGD_ERR_INVALID_ARG, std::string(path) + ": shape " + std::to_string(hdr[0]) + "x" + std::to_string(hdr[1]) +
                                                  " k=" + std::to_string(hdr[2]) + " does not match the handle");
    }
    try {
        std::vector<float> P((size_t)dims[0] * dims[2]), Q((size_t)dims[1] * dims[2]);
        const bool ok = std::fread(P.data(), sizeof(float), P.size(), f) == P.size() &&
                        std::fread(Q.data(), sizeof(float), Q.size(), f) == Q.size();
        std::fclose(f);
        if (!ok) return io_fail(MFSGD_ERR_INVALID_ARG, std::string(path) + ": truncated");
        return mfsgd_set_factors(h, P.data(), Q.data());
    } catch (const std::bad_alloc&) {
        std::fclose(f);
        return io_fail(MFSGD_ERR_OOM, "load_factors: out of host memory");
    }
}

}  // extern "C"
