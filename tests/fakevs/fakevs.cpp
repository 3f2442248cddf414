// A small VapourSynth-API-v4 host for tests: enough of VSAPI / VSPLUGINAPI to load
// libvszip.so, build clips from caller-supplied planes, invoke plugin functions with an
// argument map, and pull frames through the filter's getFrame state machine (arInitial ->
// requested frames resolved recursively -> arAllFramesReady). No VapourSynth exists in the
// build image or on the GPU box; this is what exercises the plugin boundary there.
// It is built against the same VapourSynth4_min.h as the plugin (see that header's note).
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <cmath>
#include <thread>
#include <vector>

#include "../../vapoursynth-zip_amd/plugin/VapourSynth4_min.h"
#include "../../vapoursynth-zip_amd/plugin/vsapi_layout_check.h"

struct Prop {
    int type = ptUnset;
    std::vector<int64_t> i;
    std::vector<double> f;
    std::vector<std::string> d;
    std::vector<VSNode *> nodes;
};
struct VSMap {
    std::vector<std::pair<std::string, Prop>> items;
    std::string error;
    bool has_error = false;
    Prop *find(const char *k) {
        for (auto &kv : items)
            if (kv.first == k) return &kv.second;
        return nullptr;
    }
    const Prop *find(const char *k) const { return const_cast<VSMap *>(this)->find(k); }
    Prop &get(const char *k) {
        if (Prop *p = find(k)) return *p;
        items.emplace_back(k, Prop());
        return items.back().second;
    }
};
// Plane memory comes from a size-keyed pool, like VapourSynth's frame memory pool: a steady
// pipeline recycles (already touched) buffers instead of mapping fresh pages per frame.
struct PlanePool {
    std::mutex mu;
    std::multimap<size_t, uint8_t *> idle;
    size_t idle_bytes = 0;
    bool refill = true;  // tests: recycled planes are poisoned again; the throughput tool turns it off
    uint8_t *take(size_t bytes) {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto it = idle.find(bytes);
            if (it != idle.end()) {
                uint8_t *p = it->second;
                idle.erase(it);
                idle_bytes -= bytes;
                if (refill) memset(p, 0xA5, bytes);
                return p;
            }
        }
        void *mem = nullptr;
        if (posix_memalign(&mem, 64, bytes)) abort();
        memset(mem, 0xA5, bytes);  // fresh memory only: catches reads of never-written samples
        return static_cast<uint8_t *>(mem);
    }
    void give(uint8_t *p, size_t bytes) {
        std::lock_guard<std::mutex> lk(mu);
        if (idle_bytes + bytes > ((size_t)4 << 30)) {
            free(p);
            return;
        }
        idle.emplace(bytes, p);
        idle_bytes += bytes;
    }
};
static PlanePool g_planes;
struct PlaneBuf {
    uint8_t *base = nullptr;
    size_t bytes = 0;
    ~PlaneBuf() {
        if (base) g_planes.give(base, bytes);
    }
};
struct VSFrame {
    std::atomic<int> refs{1};
    VSVideoFormat fmt;
    int w, h;
    std::shared_ptr<PlaneBuf> buf[3];
    uint8_t *ptr[3] = {nullptr, nullptr, nullptr};
    ptrdiff_t stride[3] = {0, 0, 0};
    VSMap props;
};
struct VSNode {
    std::atomic<int> refs{1};
    VSVideoInfo vi;
    std::vector<VSFrame *> frames;  // source node
    std::string name;
    VSFilterGetFrame getFrame = nullptr;
    VSFilterFree freeFn = nullptr;
    void *inst = nullptr;
    std::vector<VSNode *> deps;
};
struct VSFrameContext {
    std::vector<std::pair<VSNode *, int>> requests;
    std::map<std::pair<VSNode *, int>, const VSFrame *> ready;
    std::string error;
    bool has_error = false;
};
struct Func {
    std::string args, ret;
    VSPublicFunction fn;
    void *data;
};
struct VSPlugin {
    std::string id, ns, name;
    int version = 0;
    std::map<std::string, Func> funcs;
    void *dl = nullptr;
};
struct VSCore {
    std::vector<std::unique_ptr<VSPlugin>> plugins;
    int frame_alignment = 32;
};

static VSCore g_core;
static const VSAPI *api();

static void node_unref(VSNode *n);
static void map_clear(VSMap *m) {
    for (auto &kv : m->items)
        for (VSNode *n : kv.second.nodes) node_unref(n);
    m->items.clear();
    m->error.clear();
    m->has_error = false;
}

// ---- frames -----------------------------------------------------------------
static int plane_w(const VSVideoFormat &f, int w, int p) { return p ? w >> f.subSamplingW : w; }
static int plane_h(const VSVideoFormat &f, int h, int p) { return p ? h >> f.subSamplingH : h; }

static VSFrame *frame_new(const VSVideoFormat *fmt, int w, int h, int extra_stride = 0, int offset = 0) {
    VSFrame *f = new VSFrame();
    f->fmt = *fmt;
    f->w = w;
    f->h = h;
    const int al = g_core.frame_alignment;
    for (int p = 0; p < fmt->numPlanes; ++p) {
        const int pw = plane_w(*fmt, w, p), ph = plane_h(*fmt, h, p);
        ptrdiff_t st = ((ptrdiff_t)pw * fmt->bytesPerSample + al - 1) / al * al + extra_stride;
        f->buf[p] = std::make_shared<PlaneBuf>();
        f->buf[p]->bytes = (size_t)st * ph + offset + 64;
        f->buf[p]->base = g_planes.take(f->buf[p]->bytes);
        f->ptr[p] = f->buf[p]->base + offset;
        f->stride[p] = st;
    }
    return f;
}
static void frame_unref(const VSFrame *cf) {
    VSFrame *f = const_cast<VSFrame *>(cf);
    if (f && --f->refs == 0) {
        map_clear(&f->props);
        delete f;
    }
}
static void map_copy(const VSMap *s, VSMap *d) {
    for (auto &kv : s->items) {
        Prop &p = d->get(kv.first.c_str());
        for (VSNode *n : p.nodes) node_unref(n);
        p = kv.second;
        for (VSNode *n : p.nodes) n->refs++;
    }
}

// ---- nodes ------------------------------------------------------------------
static void node_unref(VSNode *n) {
    if (!n || --n->refs > 0) return;
    if (n->freeFn) n->freeFn(n->inst, &g_core, api());
    for (VSFrame *f : n->frames) frame_unref(f);
    delete n;
}

static const VSFrame *node_get_frame(VSNode *node, int n, std::string *err) {
    if (n < 0) n = 0;
    if (!node->getFrame) {
        VSFrame *f = node->frames[std::min<size_t>(n, node->frames.size() - 1)];
        f->refs++;
        return f;
    }
    if (n >= node->vi.numFrames) n = node->vi.numFrames - 1;
    VSFrameContext ctx;
    void *frameData = nullptr;
    const VSFrame *r = node->getFrame(n, arInitial, node->inst, &frameData, &ctx, &g_core, api());
    if (!r && !ctx.has_error) {
        for (size_t i = 0; i < ctx.requests.size(); ++i) {  // getFrame may not append while we iterate: copy
            auto rq = ctx.requests[i];
            if (ctx.ready.count(rq)) continue;
            const VSFrame *f = node_get_frame(rq.first, rq.second, err);
            if (!f) {
                for (auto &kv : ctx.ready) frame_unref(kv.second);
                return nullptr;
            }
            ctx.ready[rq] = f;
        }
        r = node->getFrame(n, arAllFramesReady, node->inst, &frameData, &ctx, &g_core, api());
    }
    for (auto &kv : ctx.ready) frame_unref(kv.second);
    if (!r) *err = ctx.has_error ? ctx.error : (node->name + ": filter returned no frame");
    return r;
}

// ---- VSAPI implementation ---------------------------------------------------------
#define A(name) api_##name
static void VS_CC A(createVideoFilter)(VSMap *out, const char *name, const VSVideoInfo *vi, VSFilterGetFrame gf, VSFilterFree fr, int, const VSFilterDependency *deps, int nd,
                                       void *inst, VSCore *) {
    VSNode *n = new VSNode();
    n->vi = *vi;
    n->name = name;
    n->getFrame = gf;
    n->freeFn = fr;
    n->inst = inst;
    for (int i = 0; i < nd; ++i) n->deps.push_back(deps[i].source);
    Prop &p = out->get("clip");
    p.type = ptVideoNode;
    p.nodes.push_back(n);
}
static void VS_CC A(freeNode)(VSNode *n) { node_unref(n); }
static VSNode *VS_CC A(addNodeRef)(VSNode *n) {
    n->refs++;
    return n;
}
static int VS_CC A(getNodeType)(VSNode *) { return mtVideo; }
static const VSVideoInfo *VS_CC A(getVideoInfo)(VSNode *n) { return &n->vi; }
static VSFrame *VS_CC A(newVideoFrame)(const VSVideoFormat *fmt, int w, int h, const VSFrame *propSrc, VSCore *) {
    VSFrame *f = frame_new(fmt, w, h);
    if (propSrc) map_copy(&propSrc->props, &f->props);
    return f;
}
static VSFrame *VS_CC A(newVideoFrame2)(const VSVideoFormat *fmt, int w, int h, const VSFrame **planeSrc, const int *planes, const VSFrame *propSrc, VSCore *) {
    VSFrame *f = frame_new(fmt, w, h);
    if (propSrc) map_copy(&propSrc->props, &f->props);
    for (int p = 0; p < fmt->numPlanes; ++p) {
        if (planeSrc && planeSrc[p]) {  // like the real core: the plane buffer is shared (copy on write)
            const VSFrame *s = planeSrc[p];
            const int sp = planes ? planes[p] : p;
            f->buf[p] = s->buf[sp];
            f->ptr[p] = s->ptr[sp];
            f->stride[p] = s->stride[sp];
        }
    }
    return f;
}
static void VS_CC A(freeFrame)(const VSFrame *f) { frame_unref(f); }
static const VSFrame *VS_CC A(addFrameRef)(const VSFrame *f) {
    const_cast<VSFrame *>(f)->refs++;
    return f;
}
// Like the real core: the copy shares the plane buffers, and a plane is duplicated only when
// someone asks for a write pointer to a buffer that is still shared (getWritePtr below).
static VSFrame *VS_CC A(copyFrame)(const VSFrame *s, VSCore *) {
    VSFrame *f = new VSFrame();
    f->fmt = s->fmt;
    f->w = s->w;
    f->h = s->h;
    map_copy(&s->props, &f->props);
    for (int p = 0; p < s->fmt.numPlanes; ++p) {
        f->buf[p] = s->buf[p];
        f->ptr[p] = s->ptr[p];
        f->stride[p] = s->stride[p];
    }
    return f;
}
static const VSMap *VS_CC A(getFramePropertiesRO)(const VSFrame *f) { return &f->props; }
static VSMap *VS_CC A(getFramePropertiesRW)(VSFrame *f) { return &f->props; }
static ptrdiff_t VS_CC A(getStride)(const VSFrame *f, int p) { return f->stride[p]; }
static const uint8_t *VS_CC A(getReadPtr)(const VSFrame *f, int p) { return f->ptr[p]; }
static uint8_t *VS_CC A(getWritePtr)(VSFrame *f, int p) {
    if (f->buf[p] && f->buf[p].use_count() > 1) {  // copy on write
        auto nb = std::make_shared<PlaneBuf>();
        nb->bytes = f->buf[p]->bytes;
        nb->base = g_planes.take(nb->bytes);
        memcpy(nb->base, f->buf[p]->base, nb->bytes);
        f->ptr[p] = nb->base + (f->ptr[p] - f->buf[p]->base);
        f->buf[p] = nb;
    }
    return f->ptr[p];
}
static const VSVideoFormat *VS_CC A(getVideoFrameFormat)(const VSFrame *f) { return &f->fmt; }
static int VS_CC A(getFrameType)(const VSFrame *) { return mtVideo; }
static int VS_CC A(getFrameWidth)(const VSFrame *f, int p) { return plane_w(f->fmt, f->w, p); }
static int VS_CC A(getFrameHeight)(const VSFrame *f, int p) { return plane_h(f->fmt, f->h, p); }
static int VS_CC A(queryVideoFormat)(VSVideoFormat *f, int cf, int st, int bits, int ssw, int ssh, VSCore *) {
    f->colorFamily = cf;
    f->sampleType = st;
    f->bitsPerSample = bits;
    f->bytesPerSample = bits <= 8 ? 1 : (bits <= 16 ? 2 : 4);
    f->subSamplingW = ssw;
    f->subSamplingH = ssh;
    f->numPlanes = cf == cfGray ? 1 : 3;
    return 1;
}
static uint32_t VS_CC A(queryVideoFormatID)(int cf, int st, int bits, int ssw, int ssh, VSCore *) { return (uint32_t)VS_MAKE_VIDEO_ID(cf, st, bits, ssw, ssh); }
static int VS_CC A(getVideoFormatByID)(VSVideoFormat *f, uint32_t id, VSCore *c) {
    return A(queryVideoFormat)(f, (id >> 28) & 0xF, (id >> 24) & 0xF, (id >> 16) & 0xFF, (id >> 8) & 0xFF, id & 0xFF, c);
}
static const VSFrame *VS_CC A(getFrame)(int n, VSNode *node, char *errorMsg, int bufSize) {
    std::string err;
    const VSFrame *f = node_get_frame(node, n, &err);
    if (!f && errorMsg && bufSize > 0) snprintf(errorMsg, bufSize, "%s", err.c_str());
    return f;
}
static const VSFrame *VS_CC A(getFrameFilter)(int n, VSNode *node, VSFrameContext *ctx) {
    if (n < 0) n = 0;
    if (n >= node->vi.numFrames) n = node->vi.numFrames - 1;
    auto it = ctx->ready.find({node, n});
    if (it == ctx->ready.end()) return nullptr;
    const_cast<VSFrame *>(it->second)->refs++;
    return it->second;
}
static void VS_CC A(requestFrameFilter)(int n, VSNode *node, VSFrameContext *ctx) {
    if (n < 0) n = 0;
    if (n >= node->vi.numFrames) n = node->vi.numFrames - 1;
    ctx->requests.push_back({node, n});
}
static void VS_CC A(setFilterError)(const char *msg, VSFrameContext *ctx) {
    ctx->error = msg;
    ctx->has_error = true;
}
static VSMap *VS_CC A(createMap)(void) { return new VSMap(); }
static void VS_CC A(freeMap)(VSMap *m) {
    if (m) {
        map_clear(m);
        delete m;
    }
}
static void VS_CC A(clearMap)(VSMap *m) { map_clear(m); }
static void VS_CC A(copyMap)(const VSMap *s, VSMap *d) { map_copy(s, d); }
static void VS_CC A(mapSetError)(VSMap *m, const char *e) {
    map_clear(m);
    m->error = e ? e : "Error: no error specified";
    m->has_error = true;
}
static const char *VS_CC A(mapGetError)(const VSMap *m) { return m->has_error ? m->error.c_str() : nullptr; }
static int VS_CC A(mapNumKeys)(const VSMap *m) { return (int)m->items.size(); }
static const char *VS_CC A(mapGetKey)(const VSMap *m, int i) { return m->items[i].first.c_str(); }
static int VS_CC A(mapDeleteKey)(VSMap *m, const char *k) {
    for (size_t i = 0; i < m->items.size(); ++i)
        if (m->items[i].first == k) {
            for (VSNode *n : m->items[i].second.nodes) node_unref(n);
            m->items.erase(m->items.begin() + i);
            return 1;
        }
    return 0;
}
static int prop_count(const Prop *p) {
    if (!p) return -1;
    switch (p->type) {
        case ptInt: return (int)p->i.size();
        case ptFloat: return (int)p->f.size();
        case ptData: return (int)p->d.size();
        case ptVideoNode: return (int)p->nodes.size();
        default: return 0;
    }
}
static int VS_CC A(mapNumElements)(const VSMap *m, const char *k) { return prop_count(m->find(k)); }
static int VS_CC A(mapGetType)(const VSMap *m, const char *k) {
    const Prop *p = m->find(k);
    return p ? p->type : ptUnset;
}
template <typename V>
static bool fetch(const VSMap *m, const char *k, int idx, int type, const std::vector<V> Prop::*vec, V *out, int *error) {
    const Prop *p = m->find(k);
    int e = peSuccess;
    if (!p)
        e = peUnset;
    else if (p->type != type)
        e = peType;
    else if (idx < 0 || idx >= (int)(p->*vec).size())
        e = peIndex;
    if (error) *error = e;
    if (e != peSuccess) return false;
    *out = (p->*vec)[idx];
    return true;
}
static int64_t VS_CC A(mapGetInt)(const VSMap *m, const char *k, int idx, int *err) {
    int64_t v = 0;
    fetch<int64_t>(m, k, idx, ptInt, &Prop::i, &v, err);
    return v;
}
static const int64_t *VS_CC A(mapGetIntArray)(const VSMap *m, const char *k, int *err) {
    const Prop *p = m->find(k);
    if (!p || p->type != ptInt) {
        if (err) *err = p ? peType : peUnset;
        return nullptr;
    }
    if (err) *err = peSuccess;
    return p->i.data();
}
static int VS_CC A(mapSetInt)(VSMap *m, const char *k, int64_t v, int append) {
    Prop &p = m->get(k);
    if (append == maReplace || p.type != ptInt) {
        p = Prop();
        p.type = ptInt;
    }
    p.i.push_back(v);
    return 0;
}
static double VS_CC A(mapGetFloat)(const VSMap *m, const char *k, int idx, int *err) {
    double v = 0;
    fetch<double>(m, k, idx, ptFloat, &Prop::f, &v, err);
    return v;
}
static const double *VS_CC A(mapGetFloatArray)(const VSMap *m, const char *k, int *err) {
    const Prop *p = m->find(k);
    if (!p || p->type != ptFloat) {
        if (err) *err = p ? peType : peUnset;
        return nullptr;
    }
    if (err) *err = peSuccess;
    return p->f.data();
}
static int VS_CC A(mapSetFloat)(VSMap *m, const char *k, double v, int append) {
    Prop &p = m->get(k);
    if (append == maReplace || p.type != ptFloat) {
        p = Prop();
        p.type = ptFloat;
    }
    p.f.push_back(v);
    return 0;
}
static const char *VS_CC A(mapGetData)(const VSMap *m, const char *k, int idx, int *err) {
    const Prop *p = m->find(k);
    int e = !p ? peUnset : (p->type != ptData ? peType : ((idx < 0 || idx >= (int)p->d.size()) ? peIndex : peSuccess));
    if (err) *err = e;
    return e == peSuccess ? p->d[idx].c_str() : nullptr;
}
static int VS_CC A(mapGetDataSize)(const VSMap *m, const char *k, int idx, int *err) {
    const char *s = A(mapGetData)(m, k, idx, err);
    return s ? (int)m->find(k)->d[idx].size() : 0;
}
static int VS_CC A(mapSetData)(VSMap *m, const char *k, const char *data, int size, int, int append) {
    Prop &p = m->get(k);
    if (append == maReplace || p.type != ptData) {
        p = Prop();
        p.type = ptData;
    }
    p.d.emplace_back(size < 0 ? std::string(data) : std::string(data, size));
    return 0;
}
static VSNode *VS_CC A(mapGetNode)(const VSMap *m, const char *k, int idx, int *err) {
    const Prop *p = m->find(k);
    int e = !p ? peUnset : (p->type != ptVideoNode ? peType : ((idx < 0 || idx >= (int)p->nodes.size()) ? peIndex : peSuccess));
    if (err) *err = e;
    if (e != peSuccess) return nullptr;
    p->nodes[idx]->refs++;
    return p->nodes[idx];
}
static int VS_CC A(mapSetNode)(VSMap *m, const char *k, VSNode *n, int append) {
    Prop &p = m->get(k);
    if (append == maReplace || p.type != ptVideoNode) {
        for (VSNode *o : p.nodes) node_unref(o);
        p = Prop();
        p.type = ptVideoNode;
    }
    n->refs++;
    p.nodes.push_back(n);
    return 0;
}
static int VS_CC A(mapConsumeNode)(VSMap *m, const char *k, VSNode *n, int append) {
    A(mapSetNode)(m, k, n, append);
    node_unref(n);
    return 0;
}
static VSPlugin *VS_CC A(getPluginByID)(const char *id, VSCore *c) {
    for (auto &p : c->plugins)
        if (p->id == id) return p.get();
    return nullptr;
}
static VSPlugin *VS_CC A(getPluginByNamespace)(const char *ns, VSCore *c) {
    for (auto &p : c->plugins)
        if (p->ns == ns) return p.get();
    return nullptr;
}

// Argument check the way the core does it before calling the function: required
// arguments present, no unknown arguments, element kinds match the signature.
static bool check_args(const std::string &fname, const std::string &sig, const VSMap *args, VSMap *out) {
    std::vector<std::string> known;
    size_t pos = 0;
    while (pos < sig.size()) {
        size_t end = sig.find(';', pos);
        if (end == std::string::npos) end = sig.size();
        const std::string item = sig.substr(pos, end - pos);
        pos = end + 1;
        if (item.empty() || item == "any") continue;
        std::vector<std::string> parts;
        size_t q = 0;
        while (true) {
            size_t c = item.find(':', q);
            parts.push_back(item.substr(q, c == std::string::npos ? std::string::npos : c - q));
            if (c == std::string::npos) break;
            q = c + 1;
        }
        const std::string &key = parts[0];
        std::string type = parts.size() > 1 ? parts[1] : "";
        bool opt = false;
        for (size_t i = 2; i < parts.size(); ++i) opt = opt || parts[i] == "opt";
        const bool arr = type.size() > 2 && type.substr(type.size() - 2) == "[]";
        if (arr) type = type.substr(0, type.size() - 2);
        known.push_back(key);
        const Prop *p = args->find(key.c_str());
        if (!p) {
            if (!opt) {
                A(mapSetError)(out, (fname + ": argument " + key + " is required").c_str());
                return false;
            }
            continue;
        }
        const int want = type == "int" ? ptInt : type == "float" ? ptFloat : type == "data" ? ptData : ptVideoNode;
        if (p->type != want && !(want == ptFloat && p->type == ptInt)) {
            A(mapSetError)(out, (fname + ": argument " + key + " is not of the correct type").c_str());
            return false;
        }
        if (!arr && prop_count(p) > 1) {
            A(mapSetError)(out, (fname + ": argument " + key + " is not of array type but more than one value was supplied").c_str());
            return false;
        }
    }
    for (auto &kv : args->items) {
        bool ok = sig.find(";any") != std::string::npos;  // e.g. std.SetFrameProps "clip:vnode;any"
        for (auto &k : known) ok = ok || k == kv.first;
        if (!ok) {
            A(mapSetError)(out, (fname + ": no argument named " + kv.first).c_str());
            return false;
        }
    }
    return true;
}

static VSMap *VS_CC A(invoke)(VSPlugin *plugin, const char *name, const VSMap *args) {
    VSMap *out = new VSMap();
    auto it = plugin->funcs.find(name);
    if (it == plugin->funcs.end()) {
        A(mapSetError)(out, (std::string("Function '") + name + "' not found in " + plugin->ns).c_str());
        return out;
    }
    if (!check_args(name, it->second.args, args, out)) return out;
    // ints given for float arguments are converted, like the core does
    VSMap conv;
    map_copy(args, &conv);
    const std::string &sig = it->second.args;
    for (auto &kv : conv.items) {
        if (kv.second.type == ptInt && sig.find(kv.first + ":float") != std::string::npos) {
            kv.second.type = ptFloat;
            for (int64_t v : kv.second.i) kv.second.f.push_back((double)v);
            kv.second.i.clear();
        }
    }
    it->second.fn(&conv, out, it->second.data, &g_core, api());
    map_clear(&conv);
    return out;
}
static int VS_CC A(getAPIVersion)(void) { return VAPOURSYNTH_API_VERSION; }
static void VS_CC A(logMessage)(int, const char *msg, VSCore *) { fprintf(stderr, "[fakevs] %s\n", msg); }

static const VSAPI *api() {
    static VSAPI a;
    static bool init = false;
    if (!init) {
        memset(&a, 0, sizeof a);  // anything the plugin does not use stays NULL and crashes loudly
        a.createVideoFilter = A(createVideoFilter);
        a.freeNode = A(freeNode);
        a.addNodeRef = A(addNodeRef);
        a.getNodeType = A(getNodeType);
        a.getVideoInfo = A(getVideoInfo);
        a.newVideoFrame = A(newVideoFrame);
        a.newVideoFrame2 = A(newVideoFrame2);
        a.freeFrame = A(freeFrame);
        a.addFrameRef = A(addFrameRef);
        a.copyFrame = A(copyFrame);
        a.getFramePropertiesRO = A(getFramePropertiesRO);
        a.getFramePropertiesRW = A(getFramePropertiesRW);
        a.getStride = A(getStride);
        a.getReadPtr = A(getReadPtr);
        a.getWritePtr = A(getWritePtr);
        a.getVideoFrameFormat = A(getVideoFrameFormat);
        a.getFrameType = A(getFrameType);
        a.getFrameWidth = A(getFrameWidth);
        a.getFrameHeight = A(getFrameHeight);
        a.queryVideoFormat = A(queryVideoFormat);
        a.queryVideoFormatID = A(queryVideoFormatID);
        a.getVideoFormatByID = A(getVideoFormatByID);
        a.getFrame = A(getFrame);
        a.getFrameFilter = A(getFrameFilter);
        a.requestFrameFilter = A(requestFrameFilter);
        a.setFilterError = A(setFilterError);
        a.createMap = A(createMap);
        a.freeMap = A(freeMap);
        a.clearMap = A(clearMap);
        a.copyMap = A(copyMap);
        a.mapSetError = A(mapSetError);
        a.mapGetError = A(mapGetError);
        a.mapNumKeys = A(mapNumKeys);
        a.mapGetKey = A(mapGetKey);
        a.mapDeleteKey = A(mapDeleteKey);
        a.mapNumElements = A(mapNumElements);
        a.mapGetType = A(mapGetType);
        a.mapGetInt = A(mapGetInt);
        a.mapGetIntArray = A(mapGetIntArray);
        a.mapSetInt = A(mapSetInt);
        a.mapGetFloat = A(mapGetFloat);
        a.mapGetFloatArray = A(mapGetFloatArray);
        a.mapSetFloat = A(mapSetFloat);
        a.mapGetData = A(mapGetData);
        a.mapGetDataSize = A(mapGetDataSize);
        a.mapSetData = A(mapSetData);
        a.mapGetNode = A(mapGetNode);
        a.mapSetNode = A(mapSetNode);
        a.mapConsumeNode = A(mapConsumeNode);
        a.getPluginByID = A(getPluginByID);
        a.getPluginByNamespace = A(getPluginByNamespace);
        a.invoke = A(invoke);
        a.getAPIVersion = A(getAPIVersion);
        a.logMessage = A(logMessage);
        init = true;
    }
    return &a;
}

// ---- VSPLUGINAPI ---------------------------------------------------------------------
static int VS_CC papi_getAPIVersion(void) { return VAPOURSYNTH_API_VERSION; }
static int VS_CC papi_configPlugin(const char *id, const char *ns, const char *name, int version, int, int, VSPlugin *p) {
    p->id = id;
    p->ns = ns;
    p->name = name;
    p->version = version;
    return 1;
}
static int VS_CC papi_registerFunction(const char *name, const char *args, const char *ret, VSPublicFunction fn, void *data, VSPlugin *p) {
    p->funcs[name] = Func{args, ret, fn, data};
    return 1;
}

// ---- C driver API for the Python test harness ----------------------------------------------
#define DRV extern "C" __attribute__((visibility("default")))

DRV int fakevs_load_plugin(const char *path, char *err, int errlen) {
    void *dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!dl) {
        snprintf(err, errlen, "%s", dlerror());
        return -1;
    }
    auto init = reinterpret_cast<VSInitPlugin>(dlsym(dl, "VapourSynthPluginInit2"));
    if (!init) {
        snprintf(err, errlen, "VapourSynthPluginInit2 not exported");
        return -2;
    }
    static const VSPLUGINAPI papi = {papi_getAPIVersion, papi_configPlugin, papi_registerFunction};
    auto p = std::make_unique<VSPlugin>();
    p->dl = dl;
    init(p.get(), &papi);
    g_core.plugins.push_back(std::move(p));
    return 0;
}
// ---------------------------------------------------------------------------
// Stand-ins for the two core functions the plugin delegates to (opt-in, tests only):
// std.SetFrameProps and resize.Point as a plain depth conversion. They are NOT zimg: Point here
// is round-to-nearest full-range scaling (limited-range YUV integer clips: a shift), evaluated
// eagerly over the whole clip. They exist so that the plugin's delegation plumbing (which
// function, which arguments, node ownership) runs in the test host; the arithmetic of the real
// resize plugin stays outside this repository's parity claims.
// ---------------------------------------------------------------------------
static std::vector<std::string> g_standin_log;  // "std.SetFrameProps _Range=1", "resize.Point format=... dither_type=none"

static VSNode *eager_clone(VSNode *src, const VSVideoFormat *fmt) {
    VSNode *n = new VSNode();
    n->vi = src->vi;
    n->vi.format = *fmt;
    n->name = "Standin";
    return n;
}

static void VS_CC standin_set_frame_props(const VSMap *in, VSMap *out, void *, VSCore *, const VSAPI *) {
    int err = 0;
    VSNode *src = A(mapGetNode)(in, "clip", 0, &err);
    VSNode *n = eager_clone(src, &src->vi.format);
    std::string log = "std.SetFrameProps";
    for (int i = 0; i < src->vi.numFrames; ++i) {
        char e[256];
        const VSFrame *f = A(getFrame)(i, src, e, sizeof e);
        VSFrame *c = A(copyFrame)(f, &g_core);
        frame_unref(f);
        for (auto &kv : in->items) {
            if (kv.first == "clip") continue;
            if (kv.second.type == ptInt) {
                A(mapSetInt)(&c->props, kv.first.c_str(), kv.second.i[0], maReplace);
                if (i == 0) log += " " + kv.first + "=" + std::to_string(kv.second.i[0]);
            }
        }
        n->frames.push_back(c);
    }
    g_standin_log.push_back(log);
    node_unref(src);
    A(mapSetNode)(out, "clip", n, maReplace);
    node_unref(n);
}

static double sample_at(const VSFrame *f, int p, int x, int y) {
    const uint8_t *row = f->ptr[p] + (ptrdiff_t)y * f->stride[p];
    if (f->fmt.sampleType == stFloat) return f->fmt.bytesPerSample == 4 ? reinterpret_cast<const float *>(row)[x] : 0.0;
    if (f->fmt.bytesPerSample == 1) return row[x];
    if (f->fmt.bytesPerSample == 2) return reinterpret_cast<const uint16_t *>(row)[x];
    return reinterpret_cast<const uint32_t *>(row)[x];
}

static void VS_CC standin_point(const VSMap *in, VSMap *out, void *, VSCore *core, const VSAPI *) {
    int err = 0;
    VSNode *src = A(mapGetNode)(in, "clip", 0, &err);
    const int64_t id = A(mapGetInt)(in, "format", 0, &err);
    VSVideoFormat fmt;
    if (err || !A(getVideoFormatByID)(&fmt, (uint32_t)id, core) || fmt.sampleType != stInteger || fmt.bytesPerSample > 2) {
        A(mapSetError)(out, "resize: stand-in supports integer 8..16 bit targets only");
        node_unref(src);
        return;
    }
    int e2 = 0;
    const char *dither = A(mapGetData)(in, "dither_type", 0, &e2);
    g_standin_log.push_back("resize.Point format=" + std::to_string(id) + (e2 ? "" : std::string(" dither_type=") + dither));
    VSNode *n = eager_clone(src, &fmt);
    const VSVideoFormat &sf = src->vi.format;
    for (int i = 0; i < src->vi.numFrames; ++i) {
        char e[256];
        const VSFrame *f = A(getFrame)(i, src, e, sizeof e);
        VSFrame *c = frame_new(&fmt, src->vi.width, src->vi.height);
        map_copy(&f->props, &c->props);
        int pe = 0;
        const bool full = A(mapGetInt)(&f->props, "_Range", 0, &pe) == 1 && !pe;
        for (int p = 0; p < fmt.numPlanes; ++p) {
            const int pw = plane_w(fmt, c->w, p), ph = plane_h(fmt, c->h, p);
            for (int y = 0; y < ph; ++y)
                for (int x = 0; x < pw; ++x) {
                    const double v = sample_at(f, p, x, y);
                    double o;
                    const double peak_out = (double)((1u << fmt.bitsPerSample) - 1);
                    if (sf.sampleType == stFloat)
                        o = v * peak_out;
                    else if (full || sf.colorFamily == cfRGB)
                        o = v * peak_out / (double)((1ull << sf.bitsPerSample) - 1);
                    else
                        o = fmt.bitsPerSample >= sf.bitsPerSample ? v * (double)(1u << (fmt.bitsPerSample - sf.bitsPerSample)) : v / (double)(1u << (sf.bitsPerSample - fmt.bitsPerSample));
                    const long q = (long)std::floor(std::min(std::max(o, 0.0), peak_out) + 0.5);
                    uint8_t *row = c->ptr[p] + (ptrdiff_t)y * c->stride[p];
                    if (fmt.bytesPerSample == 1)
                        row[x] = (uint8_t)q;
                    else
                        reinterpret_cast<uint16_t *>(row)[x] = (uint16_t)q;
                }
        }
        frame_unref(f);
        n->frames.push_back(c);
    }
    node_unref(src);
    A(mapSetNode)(out, "clip", n, maReplace);
    node_unref(n);
}

// resize.Bicubic as SSIMULACRA2's wrapper uses it (hz.toRGBS: format=RGBS, matrix_in=...; sRGBtoLinearRGB:
// transfer=8 on a clip tagged _Transfer=13) and std.SetFrameProp(prop, intval) — for RGB / Gray clips only,
// where zimg needs no resampler: integer -> float (full range for RGB, limited for Gray, _ColorRange
// overrides), Gray -> R = G = B, and the sRGB EOTF through the approximate-gamma table. The same
// restatement as oracle/vs_host.py (pinned there by the reference's goldens); TEST INFRASTRUCTURE.
static const std::vector<float> &standin_srgb_table() {
    static const std::vector<float> t = [] {
        std::vector<float> v(65537);
        const double A = 1.055010718947587, B = 0.003041282560128;
        for (int i = 0; i < 65537; ++i) {
            const double x = (double)((float)i / 65536.0f * 2.0f - 0.5f);
            const double xc = std::max(x, 0.0);  // zimg clamps negative input (oracle/vs_host.py::srgb_eotf)
            v[i] = (float)(xc < 12.92 * B ? xc / 12.92 : std::pow((xc + (A - 1.0)) / A, 2.4));
        }
        return v;
    }();
    return t;
}

static void VS_CC standin_set_frame_prop(const VSMap *in, VSMap *out, void *, VSCore *, const VSAPI *) {
    int err = 0;
    VSNode *src = A(mapGetNode)(in, "clip", 0, &err);
    const char *prop = A(mapGetData)(in, "prop", 0, &err);
    const int64_t val = A(mapGetInt)(in, "intval", 0, &err);
    VSNode *n = eager_clone(src, &src->vi.format);
    for (int i = 0; i < src->vi.numFrames; ++i) {
        char e[256];
        const VSFrame *f = A(getFrame)(i, src, e, sizeof e);
        if (!f) {  // the upstream filter failed: the stand-in is eager, so the error surfaces at create time
            A(mapSetError)(out, e);
            node_unref(src);
            node_unref(n);
            return;
        }
        VSFrame *c = A(copyFrame)(f, &g_core);
        frame_unref(f);
        A(mapSetInt)(&c->props, prop, val, maReplace);
        n->frames.push_back(c);
    }
    g_standin_log.push_back(std::string("std.SetFrameProp ") + prop + "=" + std::to_string(val));
    node_unref(src);
    A(mapSetNode)(out, "clip", n, maReplace);
    node_unref(n);
}

// ---- YUV -> RGBS for the stand-in (hz.toRGBS on a YUV clip): the restatement of oracle/vs_host.py::yuv_to_rgbs in C++
// (zimg's integer -> float conversion, Catmull-Rom chroma resampler with two interleaved FMA accumulators, horizontal
// pass first, YUV -> RGB FMA chain). TEST INFRASTRUCTURE; tests/test_gpu_plugin.py checks its output against vs_host.
static void standin_table(int src_dim, int dst_dim, double shift, std::vector<int> &left, std::vector<float> &coef) {
    auto wgt = [](double x) {
        x = std::fabs(x);  // b = 0, c = 0.5
        if (x < 1.0) return 1.0 - 2.5 * x * x + 1.5 * x * x * x;
        if (x < 2.0) return 2.0 - 4.0 * x + 2.5 * x * x - 0.5 * x * x * x;
        return 0.0;
    };
    left.assign((size_t)dst_dim, 0);
    coef.assign((size_t)dst_dim * 4, 0.0f);
    const double scale = (double)dst_dim / src_dim;
    std::vector<double> row((size_t)src_dim);
    for (int i = 0; i < dst_dim; ++i) {
        std::fill(row.begin(), row.end(), 0.0);
        const double pos = (i + 0.5) / scale + shift, begin = std::floor(pos - 2.0 + 0.5) + 0.5;
        double w[4], total = 0;
        for (int k = 0; k < 4; ++k) total += (w[k] = wgt(begin + k - pos));
        for (int k = 0; k < 4; ++k) {
            const double xp = begin + k;
            double r = xp < 0 ? -xp : (xp >= src_dim ? 2.0 * src_dim - xp : xp);
            r = std::min(std::max(r, 0.0), std::nextafter((double)src_dim, -1.0));
            row[(size_t)std::floor(r)] += w[k] / total;
        }
        int first = -1, last = 0;
        for (int j = 0; j < src_dim; ++j)
            if (row[(size_t)j] != 0.0) {
                if (first < 0) first = j;
                last = j;
            }
        (void)last;
        const int width = std::min(4, src_dim), l = std::min(first, src_dim - width);
        left[(size_t)i] = l;
        for (int k = 0; k < 4 && l + k < src_dim; ++k) coef[(size_t)i * 4 + k] = (float)row[(size_t)(l + k)];
    }
}
static inline float standin_acc(const float *c, float x0, float x1, float x2, float x3) {
    float a0 = c[0] * x0, a1 = c[1] * x1;
    a0 = std::fmaf(c[2], x2, a0);
    a1 = std::fmaf(c[3], x3, a1);
    return a0 + a1;
}
// one chroma plane (already f32) to w x h
static std::vector<float> standin_upsample(const std::vector<float> &src, int cw, int ch, int w, int h, int ssw, int ssh, int loc) {
    auto offset = [&](int ss, bool vertical) {
        if (!ss) return 0.0;
        const double edge = -((1 << ss) - 1) / 2.0;
        if (vertical) return (loc == 2 || loc == 3) ? edge : ((loc == 4 || loc == 5) ? -edge : 0.0);
        return (loc == 0 || loc == 2 || loc == 4) ? edge : 0.0;
    };
    std::vector<float> hp;
    const std::vector<float> *cur = &src;
    int curw = cw;
    std::vector<int> left;
    std::vector<float> coef;
    if (ssw) {
        standin_table(cw, w, -offset(ssw, false) / (1 << ssw), left, coef);
        hp.resize((size_t)w * ch);
        for (int y = 0; y < ch; ++y)
            for (int x = 0; x < w; ++x) {
                const float *r = src.data() + (size_t)y * cw;
                const int l = left[(size_t)x];
                hp[(size_t)y * w + x] = standin_acc(&coef[(size_t)x * 4], r[l], r[std::min(l + 1, cw - 1)], r[std::min(l + 2, cw - 1)], r[std::min(l + 3, cw - 1)]);
            }
        cur = &hp;
        curw = w;
    }
    if (!ssh) return *cur;
    standin_table(ch, h, -offset(ssh, true) / (1 << ssh), left, coef);
    std::vector<float> out((size_t)w * h);
    for (int y = 0; y < h; ++y) {
        const int l = left[(size_t)y];
        const float *r0 = cur->data() + (size_t)l * curw, *r1 = cur->data() + (size_t)std::min(l + 1, ch - 1) * curw, *r2 = cur->data() + (size_t)std::min(l + 2, ch - 1) * curw,
                    *r3 = cur->data() + (size_t)std::min(l + 3, ch - 1) * curw;
        for (int x = 0; x < w; ++x) out[(size_t)y * w + x] = standin_acc(&coef[(size_t)y * 4], r0[x], r1[x], r2[x], r3[x]);
    }
    return out;
}
static void standin_yuv_matrix(int matrix, float m[9]) {
    double kr = 0.299, kb = 0.114;
    if (matrix == 1) { kr = 0.2126; kb = 0.0722; }
    if (matrix == 9) { kr = 0.2627; kb = 0.0593; }
    const double kg = 1.0 - kr - kb, us = 1.0 / (2.0 - 2.0 * kb), vs = 1.0 / (2.0 - 2.0 * kr);
    const double a[3][3] = {{kr, kg, kb}, {-kr * us, -kg * us, (1.0 - kb) * us}, {(1.0 - kr) * vs, -kg * vs, -kb * vs}};
    const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) + a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const int r0 = j == 0 ? 1 : 0, r1 = j == 2 ? 1 : 2, c0 = i == 0 ? 1 : 0, c1 = i == 2 ? 1 : 2;
            m[3 * i + j] = (float)((((i + j) & 1) ? -1.0 : 1.0) * (a[r0][c0] * a[r1][c1] - a[r0][c1] * a[r1][c0]) / det);
        }
}

static void VS_CC standin_bicubic(const VSMap *in, VSMap *out, void *, VSCore *core, const VSAPI *) {
    int err = 0, e_fmt = 0, e_tr = 0;
    VSNode *src = A(mapGetNode)(in, "clip", 0, &err);
    const int64_t id = A(mapGetInt)(in, "format", 0, &e_fmt);
    const int64_t transfer = A(mapGetInt)(in, "transfer", 0, &e_tr);
    const VSVideoFormat &sf = src->vi.format;
    VSVideoFormat fmt = sf;
    if (!e_fmt) A(getVideoFormatByID)(&fmt, (uint32_t)id, core);
    const bool to_rgbs = !e_fmt && fmt.colorFamily == cfRGB && fmt.sampleType == stFloat && fmt.bytesPerSample == 4;
    const bool lin = !e_tr && transfer == 8;
    const bool yuv = sf.colorFamily == cfYUV && to_rgbs && !lin;
    if (yuv) {
        int e_mi = 0;
        const int64_t matrix_in = A(mapGetInt)(in, "matrix_in", 0, &e_mi);
        g_standin_log.push_back(std::string("resize.Bicubic format=RGBS matrix_in=") + std::to_string(e_mi ? -1 : (int)matrix_in));
        VSNode *n = eager_clone(src, &fmt);
        for (int i = 0; i < src->vi.numFrames; ++i) {
            char e[256];
            const VSFrame *f = A(getFrame)(i, src, e, sizeof e);
            if (!f) {
                A(mapSetError)(out, e);
                node_unref(src);
                node_unref(n);
                return;
            }
            int pe = 0, me = 0, le = 0;
            const int64_t range = A(mapGetInt)(&f->props, "_ColorRange", 0, &pe);
            const int64_t mprop = A(mapGetInt)(&f->props, "_Matrix", 0, &me);
            const int64_t loc = A(mapGetInt)(&f->props, "_ChromaLocation", 0, &le);
            const int matrix = (!me && mprop != 2) ? (int)mprop : (e_mi ? 6 : (int)matrix_in);  // a specified frame property wins over *_in
            const bool limited = sf.sampleType == stInteger && (pe ? true : range == 1);
            const int b = sf.bitsPerSample, w = src->vi.width, h = src->vi.height, ssw = sf.subSamplingW, ssh = sf.subSamplingH;
            const int cw = (w + (1 << ssw) - 1) >> ssw, ch = (h + (1 << ssh) - 1) >> ssh;
            std::vector<float> pl[3];
            for (int p = 0; p < 3; ++p) {
                const int pw = p ? cw : w, ph = p ? ch : h;
                pl[p].resize((size_t)pw * ph);
                double off = 0, rng = 1;
                if (sf.sampleType == stInteger) {
                    off = limited ? (double)((p ? 128 : 16) << (b - 8)) : (p ? (double)(1 << (b - 1)) : 0.0);
                    rng = limited ? (double)((p ? 224 : 219) << (b - 8)) : (double)((1 << b) - 1);
                }
                const float sc = (float)(1.0 / rng), so = (float)(-off / rng);
                for (int y = 0; y < ph; ++y)
                    for (int x = 0; x < pw; ++x) {
                        const float v = (float)sample_at(f, p, x, y);
                        pl[p][(size_t)y * pw + x] = sf.sampleType == stInteger ? std::fmaf(v, sc, so) : v;
                    }
            }
            for (int p = 1; p < 3; ++p)
                if (ssw || ssh) pl[p] = standin_upsample(pl[p], cw, ch, w, h, ssw, ssh, le ? 0 : (int)loc);
            float m[9];
            standin_yuv_matrix(matrix, m);
            VSFrame *c = frame_new(&fmt, w, h);
            map_copy(&f->props, &c->props);
            A(mapSetInt)(&c->props, "_Matrix", 0, maReplace);
            for (int p = 0; p < 3; ++p)
                for (int y = 0; y < h; ++y) {
                    float *row = reinterpret_cast<float *>(c->ptr[p] + (ptrdiff_t)y * c->stride[p]);
                    for (int x = 0; x < w; ++x) {
                        const size_t o = (size_t)y * w + x;
                        row[x] = std::fmaf(m[3 * p + 2], pl[2][o], std::fmaf(m[3 * p + 1], pl[1][o], m[3 * p] * pl[0][o]));
                    }
                }
            frame_unref(f);
            n->frames.push_back(c);
        }
        node_unref(src);
        A(mapSetNode)(out, "clip", n, maReplace);
        node_unref(n);
        return;
    }
    if ((sf.colorFamily != cfRGB && sf.colorFamily != cfGray) || sf.subSamplingW || sf.subSamplingH || (!to_rgbs && !lin) || (lin && !to_rgbs && !(sf.colorFamily == cfRGB && sf.sampleType == stFloat))) {
        A(mapSetError)(out, "resize: stand-in converts RGB / Gray clips to RGBS and RGBS to linear light only");
        node_unref(src);
        return;
    }
    g_standin_log.push_back(std::string("resize.Bicubic") + (to_rgbs ? " format=RGBS" : "") + (lin ? " transfer=8" : ""));
    VSNode *n = eager_clone(src, &fmt);
    const std::vector<float> &tab = standin_srgb_table();
    for (int i = 0; i < src->vi.numFrames; ++i) {
        char e[256];
        const VSFrame *f = A(getFrame)(i, src, e, sizeof e);
        if (!f) {
            A(mapSetError)(out, e);
            node_unref(src);
            node_unref(n);
            return;
        }
        VSFrame *c = frame_new(&fmt, src->vi.width, src->vi.height);
        map_copy(&f->props, &c->props);
        int pe = 0;
        const int64_t range = A(mapGetInt)(&f->props, "_ColorRange", 0, &pe);
        const bool limited = sf.sampleType == stInteger && (pe ? sf.colorFamily == cfGray : range == 1);
        const int b = sf.bitsPerSample;
        const double rng = sf.sampleType != stInteger ? 1.0 : (limited ? (double)(219 << (b - 8)) : (double)((1 << b) - 1));
        const float off = limited ? (float)(-(double)(16 << (b - 8)) / rng) : 0.0f;
        const float sc = sf.sampleType == stInteger ? (float)(1.0 / rng) : 1.0f;
        for (int p = 0; p < 3; ++p) {
            const int sp = sf.colorFamily == cfGray ? 0 : p;
            for (int y = 0; y < c->h; ++y) {
                float *row = reinterpret_cast<float *>(c->ptr[p] + (ptrdiff_t)y * c->stride[p]);
                for (int x = 0; x < c->w; ++x) {
                    float v = (float)sample_at(f, sp, x, y);
                    if (sf.sampleType == stInteger) v = std::fmaf(v, sc, off);
                    if (lin) {
                        float t = std::nearbyintf(v * 32768.0f + 16384.0f);
                        t = std::min(std::max(t, 0.0f), 65536.0f);
                        v = tab[(size_t)t];
                    }
                    row[x] = v;
                }
            }
        }
        if (lin) A(mapSetInt)(&c->props, "_Transfer", 8, maReplace);
        frame_unref(f);
        n->frames.push_back(c);
    }
    node_unref(src);
    A(mapSetNode)(out, "clip", n, maReplace);
    node_unref(n);
}

DRV void fakevs_enable_core_standins(int on) {
    auto &pl = g_core.plugins;
    for (size_t i = 0; i < pl.size();)
        if (pl[i]->id == "com.vapoursynth.std" || pl[i]->id == "com.vapoursynth.resize")
            pl.erase(pl.begin() + i);
        else
            ++i;
    g_standin_log.clear();
    if (!on) return;
    auto st = std::make_unique<VSPlugin>();
    st->id = "com.vapoursynth.std";
    st->ns = "std";
    st->funcs["SetFrameProps"] = Func{"clip:vnode;any", "clip:vnode;", standin_set_frame_props, nullptr};
    st->funcs["SetFrameProp"] = Func{"clip:vnode;prop:data;intval:int[]:opt;", "clip:vnode;", standin_set_frame_prop, nullptr};
    pl.push_back(std::move(st));
    auto rs = std::make_unique<VSPlugin>();
    rs->id = "com.vapoursynth.resize";
    rs->ns = "resize";
    rs->funcs["Point"] = Func{"clip:vnode;format:int:opt;dither_type:data:opt;", "clip:vnode;", standin_point, nullptr};
    rs->funcs["Bicubic"] = Func{"clip:vnode;format:int:opt;matrix_in:int:opt;transfer:int:opt;", "clip:vnode;", standin_bicubic, nullptr};
    pl.push_back(std::move(rs));
}
DRV int fakevs_standin_log(int i, char *buf, int len) {
    if (i < 0 || i >= (int)g_standin_log.size()) return 0;
    snprintf(buf, (size_t)len, "%s", g_standin_log[i].c_str());
    return 1;
}

DRV void fakevs_set_alignment(int bytes) { g_core.frame_alignment = bytes; }
DRV void fakevs_set_pool_refill(int on) { g_planes.refill = on != 0; }
DRV int fakevs_plugin_info(const char *ns, char *id, int idlen, int *version, int *nfuncs) {
    VSPlugin *p = A(getPluginByNamespace)(ns, &g_core);
    if (!p) return -1;
    snprintf(id, idlen, "%s", p->id.c_str());
    *version = p->version;
    *nfuncs = (int)p->funcs.size();
    return 0;
}
DRV const char *fakevs_function_args(const char *ns, const char *fn) {
    VSPlugin *p = A(getPluginByNamespace)(ns, &g_core);
    if (!p || !p->funcs.count(fn)) return nullptr;
    return p->funcs[fn].args.c_str();
}
DRV VSNode *fakevs_source(uint32_t format_id, int w, int h, int nframes, int64_t fps_num, int64_t fps_den, int extra_stride, int offset) {
    VSNode *n = new VSNode();
    A(getVideoFormatByID)(&n->vi.format, format_id, &g_core);
    n->vi.width = w;
    n->vi.height = h;
    n->vi.numFrames = nframes;
    n->vi.fpsNum = fps_num;
    n->vi.fpsDen = fps_den;
    n->name = "Source";
    for (int i = 0; i < nframes; ++i) n->frames.push_back(frame_new(&n->vi.format, w, h, extra_stride, offset));
    return n;
}
DRV VSFrame *fakevs_source_frame(VSNode *n, int i) { return n->frames[i]; }
DRV void fakevs_node_free(VSNode *n) { node_unref(n); }
DRV void fakevs_node_info(VSNode *n, int *w, int *h, int *nframes, uint32_t *fmt, int64_t *fpsn, int64_t *fpsd) {
    *w = n->vi.width;
    *h = n->vi.height;
    *nframes = n->vi.numFrames;
    const VSVideoFormat &f = n->vi.format;
    *fmt = (uint32_t)VS_MAKE_VIDEO_ID(f.colorFamily, f.sampleType, f.bitsPerSample, f.subSamplingW, f.subSamplingH);
    *fpsn = n->vi.fpsNum;
    *fpsd = n->vi.fpsDen;
}
DRV VSMap *fakevs_map_new() { return new VSMap(); }
DRV void fakevs_map_free(VSMap *m) { A(freeMap)(m); }
DRV void fakevs_map_set_int(VSMap *m, const char *k, int64_t v) { A(mapSetInt)(m, k, v, maAppend); }
DRV void fakevs_map_set_float(VSMap *m, const char *k, double v) { A(mapSetFloat)(m, k, v, maAppend); }
DRV void fakevs_map_set_data(VSMap *m, const char *k, const char *v) { A(mapSetData)(m, k, v, -1, dtUtf8, maAppend); }
DRV void fakevs_map_set_node(VSMap *m, const char *k, VSNode *n) { A(mapSetNode)(m, k, n, maAppend); }
DRV void fakevs_map_set_empty(VSMap *m, const char *k, int type) { m->get(k).type = type; }
DRV VSMap *fakevs_invoke(const char *ns, const char *fn, VSMap *args) {
    VSPlugin *p = A(getPluginByNamespace)(ns, &g_core);
    if (!p) {
        VSMap *o = new VSMap();
        A(mapSetError)(o, "no such plugin");
        return o;
    }
    return A(invoke)(p, fn, args);
}
DRV const char *fakevs_map_error(VSMap *m) { return A(mapGetError)(m); }
DRV VSNode *fakevs_map_node(VSMap *m, const char *k) {
    int e = 0;
    return A(mapGetNode)(m, k, 0, &e);
}
DRV const VSFrame *fakevs_get_frame(VSNode *n, int i, char *err, int errlen) { return A(getFrame)(i, n, err, errlen); }
DRV void fakevs_frame_free(const VSFrame *f) { frame_unref(f); }
// What a VapourSynth output loop does under fmParallel: `threads` workers pull `count` frames
// (frame numbers first, first+1, ... modulo the clip length) and drop them. Returns the number of
// failed frames; *seconds is the wall time of the whole pull.
DRV int fakevs_pull_warm(VSNode *n, int first, int count, int threads, int warm_per_thread, double *seconds);
DRV int fakevs_pull(VSNode *n, int first, int count, int threads, double *seconds) { return fakevs_pull_warm(n, first, count, threads, 0, seconds); }
// Same with a steady-state clock: VapourSynth's workers live as long as the core, so a filter's
// per-thread state (here: the plugin's per-thread GPU contexts) is set up once per session. Every
// worker first fetches `warm_per_thread` frames untimed, the workers meet, and the clock runs from
// that point until the last timed frame is done (before the workers exit).
DRV int fakevs_pull_warm(VSNode *n, int first, int count, int threads, int warm_per_thread, double *seconds) {
    std::atomic<int> next{0}, failed{0}, arrived{0}, finished{0};
    std::atomic<int64_t> t_begin{0}, t_end{0};
    const int nframes = n->vi.numFrames;
    auto now_ns = [] { return (int64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    auto fetch = [&](int k, char *err, int errlen) {
        const VSFrame *f = A(getFrame)((first + k) % nframes, n, err, errlen);
        if (f)
            frame_unref(f);
        else
            failed.fetch_add(1);
    };
    auto work = [&](int tid) {
        char err[256];
        for (int w = 0; w < warm_per_thread; ++w) fetch(tid * warm_per_thread + w, err, (int)sizeof(err));
        if (arrived.fetch_add(1) + 1 == threads) t_begin.store(now_ns());
        while (t_begin.load() == 0) std::this_thread::yield();
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= count) break;
            fetch(k, err, (int)sizeof(err));
        }
        if (finished.fetch_add(1) + 1 == threads) t_end.store(now_ns());
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
    for (auto &t : pool) t.join();
    if (seconds) *seconds = (double)(t_end.load() - t_begin.load()) * 1e-9;
    return failed.load();
}
DRV uint8_t *fakevs_frame_plane(const VSFrame *f, int p, int *w, int *h, ptrdiff_t *stride) {
    *w = plane_w(f->fmt, f->w, p);
    *h = plane_h(f->fmt, f->h, p);
    *stride = f->stride[p];
    return f->ptr[p];
}
DRV int fakevs_frame_num_planes(const VSFrame *f) { return f->fmt.numPlanes; }
DRV int fakevs_frame_prop_count(const VSFrame *f, const char *k) { return prop_count(f->props.find(k)); }
DRV int fakevs_frame_prop_type(const VSFrame *f, const char *k) { return A(mapGetType)(&f->props, k); }
DRV int64_t fakevs_frame_prop_int(const VSFrame *f, const char *k, int i) { return A(mapGetInt)(&f->props, k, i, nullptr); }
DRV double fakevs_frame_prop_float(const VSFrame *f, const char *k, int i) { return A(mapGetFloat)(&f->props, k, i, nullptr); }
DRV void fakevs_frame_set_prop_int(VSFrame *f, const char *k, int64_t v) { A(mapSetInt)(&f->props, k, v, maReplace); }
DRV int fakevs_frame_num_props(const VSFrame *f) { return (int)f->props.items.size(); }
DRV const char *fakevs_frame_prop_key(const VSFrame *f, int i) { return f->props.items[i].first.c_str(); }
