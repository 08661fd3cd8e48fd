// rosbag_reader.cc -- see rosbag_reader.h.  Bag format 2.0: the file is "#ROSBAG V2.0\n" followed by
// records { uint32 header_len; header; uint32 data_len; data }, a header being a list of
// { uint32 field_len; "name=value" } with binary little-endian values.  Record kinds (field "op"):
// 0x03 bag header, 0x05 chunk (holds 0x07 connection and 0x02 message-data records, possibly
// compressed), 0x04 index data, 0x06 chunk info, 0x07 connection.
#include "rosbag_reader.h"

#include <dlfcn.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>

namespace gpscal_host {
namespace {

struct Span {
    const uint8_t *p;
    size_t n;
};

uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

// header fields "name=value" -> map
bool parse_header(Span h, std::map<std::string, Span> &f)
{
    size_t at = 0;
    while (at < h.n) {
        if (at + 4 > h.n) return false;
        const uint32_t len = rd32(h.p + at);
        at += 4;
        if (len == 0 || at + len > h.n) return false;
        const uint8_t *eq = (const uint8_t *)memchr(h.p + at, '=', len);
        if (!eq) return false;
        f[std::string((const char *)h.p + at, eq - (h.p + at))] = Span{eq + 1, (size_t)(h.p + at + len - (eq + 1))};
        at += len;
    }
    return true;
}

struct Msg {
    uint64_t time;  // record time, secs << 32 | nsecs
    size_t order;
    double stamp;
    size_t first_point, npoints;
};

struct Reader {
    int pass = 0;  // 0: connection records only (a message may precede its connection record in file order when the
                   // chunks are not in time order; rosbag itself reads the connection list from the index section),
                   // 1: message data
    std::string want;  // topic without leading slash
    std::map<uint32_t, bool> conn_wanted;
    std::vector<Msg> msgs;
    std::vector<float> pts;
    std::string err;

    static std::string strip(const std::string &t) { return !t.empty() && t[0] == '/' ? t.substr(1) : t; }

    bool cloud(Span d, Msg &m)
    {
        // sensor_msgs/PointCloud2, ROS1 serialisation (little endian)
        size_t at = 0;
        auto need = [&](size_t k) { return at + k <= d.n; };
        auto u32 = [&](uint32_t &v) {
            if (!need(4)) return false;
            v = rd32(d.p + at);
            at += 4;
            return true;
        };
        uint32_t seq, secs, nsecs, len, height, width, nfields;
        if (!u32(seq) || !u32(secs) || !u32(nsecs) || !u32(len) || !need(len)) return false;
        at += len;  // frame_id
        if (!u32(height) || !u32(width) || !u32(nfields)) return false;
        int off[3] = {-1, -1, -1};
        for (uint32_t k = 0; k < nfields; ++k) {
            uint32_t nl, foff, cnt;
            if (!u32(nl) || !need(nl)) return false;
            const std::string name((const char *)d.p + at, nl);
            at += nl;
            if (!u32(foff) || !need(1)) return false;
            const uint8_t dtype = d.p[at++];
            if (!u32(cnt)) return false;
            const int a = name == "x" ? 0 : name == "y" ? 1 : name == "z" ? 2 : -1;
            if (a >= 0) {
                if (dtype != 7) {  // pcl::fromROSMsg maps a field only onto the same datatype (FLOAT32)
                    err = "PointCloud2 field " + name + " is not FLOAT32";
                    return false;
                }
                off[a] = (int)foff;
            }
        }
        if (!need(1)) return false;
        const uint8_t bigendian = d.p[at++];
        uint32_t point_step, row_step, dlen;
        if (!u32(point_step) || !u32(row_step) || !u32(dlen) || !need(dlen)) return false;
        if (bigendian) {
            err = "big-endian PointCloud2 is not supported";
            return false;
        }
        if (off[0] < 0 || off[1] < 0 || off[2] < 0) {
            err = "PointCloud2 without x / y / z fields";
            return false;
        }
        const uint8_t *data = d.p + at;
        // the layout the message declares must fit its own data block: every field inside a point, every point
        // inside its row, every row inside the data
        if (point_step == 0 || (width > 0 && (size_t)width * point_step > row_step))
            return fail_cloud("PointCloud2 with point_step / row_step that cannot hold a row");
        for (int a = 0; a < 3; ++a)
            if ((size_t)off[a] + 4 > point_step) return fail_cloud("PointCloud2 field offset beyond point_step");
        if ((uint64_t)height * row_step > dlen) return fail_cloud("PointCloud2 data shorter than height x row_step");
        const size_t npts = (size_t)height * width;
        m.stamp = (double)secs + 1e-9 * (double)nsecs;  // ros::Time::toSec
        m.first_point = pts.size() / 3;
        m.npoints = npts;
        pts.resize(pts.size() + 3 * npts);
        float *o = pts.data() + 3 * m.first_point;
        for (uint32_t r = 0; r < height; ++r)
            for (uint32_t c = 0; c < width; ++c) {
                const size_t base = (size_t)r * row_step + (size_t)c * point_step;
                if (base + point_step > dlen) {
                    err = "PointCloud2 data shorter than height x width x point_step";
                    return false;
                }
                for (int a = 0; a < 3; ++a) memcpy(o++, data + base + off[a], 4);
            }
        return true;
    }

    bool fail_cloud(const char *what)
    {
        err = what;
        return false;
    }

    // records of a chunk body or of the top level
    bool records(Span s, bool top)
    {
        size_t at = 0;
        while (at < s.n) {
            if (at + 4 > s.n) return fail("truncated record");
            const uint32_t hl = rd32(s.p + at);
            if (at + 4 + hl + 4 > s.n) return fail("truncated record header");
            Span h{s.p + at + 4, hl};
            const uint32_t dl = rd32(s.p + at + 4 + hl);
            if (at + 8 + hl + (size_t)dl > s.n) return fail("truncated record data");
            Span d{s.p + at + 8 + hl, dl};
            at += 8 + (size_t)hl + dl;
            std::map<std::string, Span> f;
            if (!parse_header(h, f) || !f.count("op") || f["op"].n != 1) return fail("bad record header");
            const uint8_t op = f["op"].p[0];
            if (op == 0x07) {  // connection
                if (pass != 0) continue;
                if (!f.count("conn") || f["conn"].n != 4 || !f.count("topic")) return fail("bad connection record");
                std::map<std::string, Span> ch;
                if (!parse_header(d, ch)) return fail("bad connection header");
                const std::string topic((const char *)f["topic"].p, f["topic"].n);
                std::string type;
                if (ch.count("type")) type.assign((const char *)ch["type"].p, ch["type"].n);
                conn_wanted[rd32(f["conn"].p)] = strip(topic) == want && type == "sensor_msgs/PointCloud2";
            } else if (op == 0x02) {  // message data
                if (pass != 1) continue;
                if (!f.count("conn") || f["conn"].n != 4 || !f.count("time") || f["time"].n != 8) return fail("bad message record");
                const uint32_t c = rd32(f["conn"].p);
                if (conn_wanted.count(c) && conn_wanted[c]) {
                    Msg m;
                    m.time = (uint64_t)rd32(f["time"].p) << 32 | rd32(f["time"].p + 4);
                    m.order = msgs.size();
                    if (!cloud(d, m)) return fail(err.empty() ? "malformed PointCloud2" : err.c_str());
                    msgs.push_back(m);
                }
            } else if (op == 0x05 && top) {  // chunk
                if (!f.count("compression") || !f.count("size") || f["size"].n != 4) return fail("bad chunk record");
                const std::string comp((const char *)f["compression"].p, f["compression"].n);
                const uint32_t usize = rd32(f["size"].p);
                // rosbag's chunks are ~768 KiB (at most a few MiB): a size field beyond this is a corrupt file,
                // not a reason to allocate gigabytes
                if (usize > (256u << 20)) return fail("chunk size field is implausible (> 256 MiB)");
                if (comp == "none") {
                    if (!records(d, false)) return false;
                } else if (comp == "bz2") {
                    std::vector<uint8_t> buf(usize);
                    if (!bunzip(d, buf)) return false;
                    if (!records(Span{buf.data(), buf.size()}, false)) return false;
                } else if (comp == "lz4") {
                    std::vector<uint8_t> buf(usize);
                    if (!unlz4(d, buf)) return false;
                    if (!records(Span{buf.data(), buf.size()}, false)) return false;
                } else {
                    return fail(("chunk compression '" + comp + "' is not supported").c_str());
                }
            }
            // 0x03 bag header, 0x04 index data, 0x06 chunk info: nothing to do for a sequential read
        }
        return true;
    }

    bool bunzip(Span d, std::vector<uint8_t> &out)
    {
        typedef int (*fn_t)(char *, unsigned *, char *, unsigned, int, int);
        static fn_t fn = nullptr;
        if (!fn) {
            void *h = dlopen("libbz2.so.1.0", RTLD_NOW);
            if (!h) h = dlopen("libbz2.so.1", RTLD_NOW);
            if (h) fn = (fn_t)dlsym(h, "BZ2_bzBuffToBuffDecompress");
        }
        if (!fn) return fail("bz2 chunk but libbz2 is not available");
        unsigned n = (unsigned)out.size();
        if (fn((char *)out.data(), &n, (char *)d.p, (unsigned)d.n, 0, 0) != 0 || n != out.size())
            return fail("bz2 chunk does not decompress to its declared size");
        return true;
    }

    // roslz4 writes standard LZ4 frames; liblz4's frame API is bound at run time like libbz2
    bool unlz4(Span d, std::vector<uint8_t> &out)
    {
        typedef size_t (*create_t)(void **, unsigned);
        typedef size_t (*decomp_t)(void *, void *, size_t *, const void *, size_t *, const void *);
        typedef size_t (*free_t)(void *);
        typedef unsigned (*iserr_t)(size_t);
        static create_t f_create = nullptr;
        static decomp_t f_decomp = nullptr;
        static free_t f_free = nullptr;
        static iserr_t f_iserr = nullptr;
        if (!f_create) {
            void *h = dlopen("liblz4.so.1", RTLD_NOW);
            if (h) {
                f_create = (create_t)dlsym(h, "LZ4F_createDecompressionContext");
                f_decomp = (decomp_t)dlsym(h, "LZ4F_decompress");
                f_free = (free_t)dlsym(h, "LZ4F_freeDecompressionContext");
                f_iserr = (iserr_t)dlsym(h, "LZ4F_isError");
            }
        }
        if (!f_create || !f_decomp || !f_free || !f_iserr) return fail("lz4 chunk but liblz4 is not available");
        void *ctx = nullptr;
        if (f_iserr(f_create(&ctx, 100 /* LZ4F_VERSION */))) return fail("lz4: cannot create a decompression context");
        size_t in_at = 0, out_at = 0;
        bool ok = true;
        while (in_at < d.n) {
            size_t dst = out.size() - out_at, src = d.n - in_at;
            const size_t rc = f_decomp(ctx, out.data() + out_at, &dst, d.p + in_at, &src, nullptr);
            if (f_iserr(rc) || (dst == 0 && src == 0)) {
                ok = false;
                break;
            }
            in_at += src;
            out_at += dst;
            if (rc == 0) break;  // end of frame
        }
        f_free(ctx);
        if (!ok || out_at != out.size()) return fail("lz4 chunk does not decompress to its declared size");
        return true;
    }

    bool fail(const char *what)
    {
        if (err.empty() || err != what) err = what;
        return false;
    }
};

}  // namespace

bool read_bag_clouds(const std::string &path, const std::string &topic, CloudSeries &out, std::string &err)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        err = "open " + path + " error";
        return false;
    }
    std::vector<uint8_t> file;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    file.resize(sz > 0 ? (size_t)sz : 0);
    const bool ok = sz > 0 && fread(file.data(), 1, file.size(), f) == file.size();
    fclose(f);
    static const char magic[] = "#ROSBAG V2.0\n";
    if (!ok || file.size() < 13 || memcmp(file.data(), magic, 13) != 0) {
        err = path + " is not a rosbag V2.0 file";
        return false;
    }
    Reader R;
    R.want = Reader::strip(topic);
    for (R.pass = 0; R.pass < 2; ++R.pass)
        if (!R.records(Span{file.data() + 13, file.size() - 13}, true)) {
            err = path + ": " + R.err;
            return false;
        }
    // rosbag::View hands messages out by time; equal times keep file order
    std::stable_sort(R.msgs.begin(), R.msgs.end(), [](const Msg &a, const Msg &b) { return a.time < b.time; });
    if (out.sweep_off.empty()) out.sweep_off.push_back(0);
    for (const Msg &m : R.msgs) {
        out.xyz.insert(out.xyz.end(), R.pts.begin() + 3 * m.first_point, R.pts.begin() + 3 * (m.first_point + m.npoints));
        out.sweep_off.push_back(out.sweep_off.back() + (int)m.npoints);
        out.stamps.push_back(m.stamp);
    }
    return true;
}

}  // namespace gpscal_host
