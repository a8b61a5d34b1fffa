// The reference's checkpoint file (2d/src/bin/gui.rs:24-45): `CheckPoint { world: World, raw_image: RawImage }`
// written with rmp_serde::encode::to_vec, i.e. MessagePack in rmp-serde's compact form.  rmp-serde (1.3.1 in the
// reference's Cargo.lock) is not part of the reference tree; what follows restates its published encoding of the
// serde data model for the types involved:
//   struct / tuple struct        -> array of the fields in declaration order (no field names)
//   Vec<T>, serialize_seq         -> array
//   enum, unit variant            -> str "Variant"
//   enum, newtype variant         -> map of one entry { "Variant": value }
//   enum, struct / tuple variant  -> map of one entry { "Variant": array of fields }
//   Option<T>                     -> nil | value
//   f64 / f32                     -> float 64 / float 32;  integers -> the shortest MessagePack integer that fits
//   glam DVec2 / Vec3             -> array of 2 f64 / 3 f32 (glam's serde feature serialises vectors as tuples)
// The types (paths relative to the reference root):
//   World { camera, objects, quadtree }                 2d/src/rt_common.rs:541-546
//   Camera { center, size }                             2d/src/rt_common.rs:480-484
//   Object { shape, mat }                               2d/src/rt_common.rs:486-490
//   Shape::{Circle(Circle), Segment(Segment), Strip(Strip)}   2d/src/rt_common.rs:327-333
//   Circle { center, r }, Segment { a, b, n }, Strip { strip }   2d/src/rt_common.rs:60-64, 201-206, 283-286
//   Material::{Emissive{emission}, DirectionalEmissive{emission, angle, d}, Diffuse{absorption}, Reflective,
//              Dielectric{ior, absorption}}             2d/src/rt_common.rs:403-431
//   Ior::{Simple(f64), Cauchy{a, b}}                    2d/src/rt_common.rs:433-441
//   Spectrum                                            a seq of its 95 f32 (hand-written impl, 2d/src/spectrum.rs:125-136)
//   Node { objects, aabb, top_right, top_left, bottom_left, bottom_right }, Aabb { center, half_size }
//                                                       2d/src/aabb.rs:10-14, 111-119
//   RawImage { width, height, data }, PixelData { weight, value }   2d/src/img.rs:20-24, 70-75
// World::new (rt_common.rs:548-561) stores an EMPTY quadtree over the 800 x 600 box and the renderer never reads it
// (use_quadtree = false), so the loader skips whatever tree a file holds and the writer emits that empty root.
// Reading is lenient where encoders may differ: any integer width, f32 or f64 for any float, and the variant
// tag as a name or as its index.  The reference itself cannot read these files back (Spectrum's Deserialize is
// `todo!()`, spectrum.rs:117-124); no checkpoint file ships with it, so this codec is checked against the format
// rules above, not against a file the reference wrote.
#pragma once

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt2d.hpp"

namespace rt2d_host {
namespace checkpoint {

class FormatError : public std::runtime_error {
public:
    using std::runtime_error::runtime_error;
};

// ---- MessagePack, the subset serde's data model produces ----------------------
class Reader {
public:
    Reader(const uint8_t *data, size_t size) : p_(data), end_(data + size), begin_(data) {}
    size_t offset() const { return (size_t)(p_ - begin_); }
    bool at_end() const { return p_ == end_; }

    bool peek_nil() const { return p_ < end_ && *p_ == 0xc0; }
    bool peek_str() const { return p_ < end_ && ((*p_ >= 0xa0 && *p_ <= 0xbf) || *p_ == 0xd9 || *p_ == 0xda || *p_ == 0xdb); }
    bool peek_map() const { return p_ < end_ && ((*p_ >= 0x80 && *p_ <= 0x8f) || *p_ == 0xde || *p_ == 0xdf); }
    bool peek_array() const { return p_ < end_ && ((*p_ >= 0x90 && *p_ <= 0x9f) || *p_ == 0xdc || *p_ == 0xdd); }

    void read_nil() { if (byte() != 0xc0) fail("expected nil"); }
    bool read_bool()
    {
        const uint8_t b = byte();
        if (b == 0xc2) return false;
        if (b == 0xc3) return true;
        fail("expected bool");
    }
    int64_t read_int()
    {
        const uint8_t b = byte();
        if (b <= 0x7f) return b;
        if (b >= 0xe0) return (int8_t)b;
        switch (b) {
        case 0xcc: return (int64_t)be(1);
        case 0xcd: return (int64_t)be(2);
        case 0xce: return (int64_t)be(4);
        case 0xcf: { const uint64_t v = be(8); if (v > (uint64_t)INT64_MAX) fail("integer out of range"); return (int64_t)v; }
        case 0xd0: return (int8_t)be(1);
        case 0xd1: return (int16_t)be(2);
        case 0xd2: return (int32_t)be(4);
        case 0xd3: return (int64_t)be(8);
        default: fail("expected integer");
        }
    }
    double read_float()
    {
        if (p_ >= end_) fail("truncated");
        if (*p_ == 0xca) { ++p_; const uint32_t u = (uint32_t)be(4); float f; memcpy(&f, &u, 4); return (double)f; }
        if (*p_ == 0xcb) { ++p_; const uint64_t u = be(8); double d; memcpy(&d, &u, 8); return d; }
        return (double)read_int(); // an encoder may shorten integral values
    }
    std::string read_str()
    {
        const uint8_t b = byte();
        size_t n;
        if (b >= 0xa0 && b <= 0xbf) n = b & 0x1f;
        else if (b == 0xd9) n = (size_t)be(1);
        else if (b == 0xda) n = (size_t)be(2);
        else if (b == 0xdb) n = (size_t)be(4);
        else fail("expected str");
        need(n);
        std::string s((const char *)p_, n);
        p_ += n;
        return s;
    }
    size_t read_array_len()
    {
        const uint8_t b = byte();
        if (b >= 0x90 && b <= 0x9f) return b & 0x0f;
        if (b == 0xdc) return (size_t)be(2);
        if (b == 0xdd) return (size_t)be(4);
        fail("expected array");
    }
    size_t read_map_len()
    {
        const uint8_t b = byte();
        if (b >= 0x80 && b <= 0x8f) return b & 0x0f;
        if (b == 0xde) return (size_t)be(2);
        if (b == 0xdf) return (size_t)be(4);
        fail("expected map");
    }
    void expect_array(size_t n, const char *what)
    {
        const size_t got = read_array_len();
        if (got != n) fail(std::string(what) + ": expected " + std::to_string(n) + " fields, found " + std::to_string(got));
    }
    // Skips one value of any type (iteratively: a quadtree can be deep).
    void skip()
    {
        size_t pending = 1;
        while (pending) {
            --pending;
            const uint8_t b = byte();
            if (b <= 0x7f || b >= 0xe0 || b == 0xc0 || b == 0xc2 || b == 0xc3) continue;
            if (b >= 0xa0 && b <= 0xbf) { advance(b & 0x1f); continue; }
            if (b >= 0x90 && b <= 0x9f) { pending += b & 0x0f; continue; }
            if (b >= 0x80 && b <= 0x8f) { pending += 2 * (size_t)(b & 0x0f); continue; }
            switch (b) {
            case 0xc4: case 0xd9: advance((size_t)be(1)); break;
            case 0xc5: case 0xda: advance((size_t)be(2)); break;
            case 0xc6: case 0xdb: advance((size_t)be(4)); break;
            case 0xc7: advance((size_t)be(1) + 1); break;
            case 0xc8: advance((size_t)be(2) + 1); break;
            case 0xc9: advance((size_t)be(4) + 1); break;
            case 0xca: case 0xce: case 0xd2: advance(4); break;
            case 0xcb: case 0xcf: case 0xd3: advance(8); break;
            case 0xcc: case 0xd0: advance(1); break;
            case 0xcd: case 0xd1: advance(2); break;
            case 0xd4: advance(2); break;
            case 0xd5: advance(3); break;
            case 0xd6: advance(5); break;
            case 0xd7: advance(9); break;
            case 0xd8: advance(17); break;
            case 0xdc: pending += (size_t)be(2); break;
            case 0xdd: pending += (size_t)be(4); break;
            case 0xde: pending += 2 * (size_t)be(2); break;
            case 0xdf: pending += 2 * (size_t)be(4); break;
            default: fail("reserved type byte 0xc1");
            }
        }
    }
    [[noreturn]] void fail(const std::string &what) const
    {
        throw FormatError("checkpoint: " + what + " at byte " + std::to_string(offset()));
    }

private:
    void need(size_t n) const { if ((size_t)(end_ - p_) < n) fail("truncated"); }
    void advance(size_t n) { need(n); p_ += n; }
    uint8_t byte() { need(1); return *p_++; }
    uint64_t be(int n)
    {
        need((size_t)n);
        uint64_t v = 0;
        for (int i = 0; i < n; ++i) v = (v << 8) | *p_++;
        return v;
    }
    const uint8_t *p_, *end_, *begin_;
};

class Writer {
public:
    std::vector<uint8_t> out;
    void nil() { out.push_back(0xc0); }
    void uint(uint64_t v)
    {
        if (v <= 0x7f) out.push_back((uint8_t)v);
        else if (v <= 0xff) { out.push_back(0xcc); be(v, 1); }
        else if (v <= 0xffff) { out.push_back(0xcd); be(v, 2); }
        else if (v <= 0xffffffffull) { out.push_back(0xce); be(v, 4); }
        else { out.push_back(0xcf); be(v, 8); }
    }
    void sint(int64_t v) // rmp's write_sint: non-negative values take the unsigned forms
    {
        if (v >= 0) { uint((uint64_t)v); return; }
        if (v >= -32) out.push_back((uint8_t)v);
        else if (v >= INT8_MIN) { out.push_back(0xd0); be((uint64_t)v, 1); }
        else if (v >= INT16_MIN) { out.push_back(0xd1); be((uint64_t)v, 2); }
        else if (v >= INT32_MIN) { out.push_back(0xd2); be((uint64_t)v, 4); }
        else { out.push_back(0xd3); be((uint64_t)v, 8); }
    }
    void f32(float v) { uint32_t u; memcpy(&u, &v, 4); out.push_back(0xca); be(u, 4); }
    void f64(double v) { uint64_t u; memcpy(&u, &v, 8); out.push_back(0xcb); be(u, 8); }
    void str(const char *s)
    {
        const size_t n = strlen(s);
        if (n <= 31) out.push_back((uint8_t)(0xa0 | n));
        else if (n <= 0xff) { out.push_back(0xd9); be(n, 1); }
        else { out.push_back(0xda); be(n, 2); }
        out.insert(out.end(), s, s + n);
    }
    void array(size_t n)
    {
        if (n <= 15) out.push_back((uint8_t)(0x90 | n));
        else if (n <= 0xffff) { out.push_back(0xdc); be(n, 2); }
        else { out.push_back(0xdd); be(n, 4); }
    }
    void map(size_t n)
    {
        if (n <= 15) out.push_back((uint8_t)(0x80 | n));
        else if (n <= 0xffff) { out.push_back(0xde); be(n, 2); }
        else { out.push_back(0xdf); be(n, 4); }
    }
    void variant(const char *name) { map(1); str(name); } // followed by the payload

private:
    void be(uint64_t v, int n) { for (int i = n - 1; i >= 0; --i) out.push_back((uint8_t)(v >> (8 * i))); }
};

// ---- the reference's types ------------------------------------------------------
inline void put(Writer &w, DVec2 v) { w.array(2); w.f64(v.x); w.f64(v.y); }
inline DVec2 get_dvec2(Reader &r) { r.expect_array(2, "DVec2"); DVec2 v; v.x = r.read_float(); v.y = r.read_float(); return v; }

inline void put(Writer &w, const Spectrum &s) { w.array(s.data.size()); for (float v : s.data) w.f32(v); }
inline Spectrum get_spectrum(Reader &r)
{
    r.expect_array(SPECTRUM_SAMPLES, "Spectrum");
    Spectrum s;
    for (float &v : s.data) v = (float)r.read_float();
    return s;
}

inline void put(Writer &w, const Segment &s) { w.array(3); put(w, s.a); put(w, s.b); put(w, s.n); }
inline Segment get_segment(Reader &r)
{
    r.expect_array(3, "Segment");
    Segment s; // the stored normal is kept as written, not recomputed
    s.a = get_dvec2(r); s.b = get_dvec2(r); s.n = get_dvec2(r);
    return s;
}

// An enum value: (variant, has_payload).  `names` in declaration order; a unit variant is a bare tag.
inline int get_variant(Reader &r, const char *const *names, int n_names, const char *what, bool &has_payload)
{
    auto by_tag = [&](Reader &rr) -> int {
        if (rr.peek_str()) {
            const std::string tag = rr.read_str();
            for (int i = 0; i < n_names; ++i)
                if (tag == names[i]) return i;
            rr.fail(std::string(what) + ": unknown variant '" + tag + "'");
        }
        const int64_t idx = rr.read_int();
        if (idx < 0 || idx >= n_names) rr.fail(std::string(what) + ": variant index out of range");
        return (int)idx;
    };
    if (r.peek_map()) {
        if (r.read_map_len() != 1) r.fail(std::string(what) + ": an enum is a map of one entry");
        has_payload = true;
        return by_tag(r);
    }
    if (r.peek_array()) { // older encoders: [variant index, payload]
        if (r.read_array_len() != 2) r.fail(std::string(what) + ": an enum is [variant, payload]");
        has_payload = true;
        return by_tag(r);
    }
    has_payload = false;
    return by_tag(r);
}

inline void put(Writer &w, const Shape &s)
{
    if (const Circle *c = std::get_if<Circle>(&s)) {
        w.variant("Circle"); w.array(2); put(w, c->center); w.f64(c->r);
    } else if (const Segment *g = std::get_if<Segment>(&s)) {
        w.variant("Segment"); put(w, *g);
    } else {
        const Strip &t = std::get<Strip>(s);
        w.variant("Strip"); w.array(1); w.array(t.strip.size());
        for (const Segment &g2 : t.strip) put(w, g2);
    }
}
inline Shape get_shape(Reader &r)
{
    static const char *const names[] = { "Circle", "Segment", "Strip" };
    bool payload;
    const int v = get_variant(r, names, 3, "Shape", payload);
    if (!payload) r.fail("Shape: variant without data");
    if (v == 0) {
        r.expect_array(2, "Circle");
        Circle c; c.center = get_dvec2(r); c.r = r.read_float();
        return c;
    }
    if (v == 1) return get_segment(r);
    r.expect_array(1, "Strip");
    Strip t;
    const size_t n = r.read_array_len();
    t.strip.reserve(n);
    for (size_t i = 0; i < n; ++i) t.strip.push_back(get_segment(r));
    return t;
}

inline void put(Writer &w, const Ior &i)
{
    if (i.kind == Ior::Simple) { w.variant("Simple"); w.f64(i.a); }
    else { w.variant("Cauchy"); w.array(2); w.f64(i.a); w.f64(i.b); }
}
inline Ior get_ior(Reader &r)
{
    static const char *const names[] = { "Simple", "Cauchy" };
    bool payload;
    const int v = get_variant(r, names, 2, "Ior", payload);
    if (!payload) r.fail("Ior: variant without data");
    if (v == 0) return Ior::simple(r.read_float());
    r.expect_array(2, "Ior::Cauchy");
    const double a = r.read_float(), b = r.read_float();
    return Ior::cauchy(a, b);
}

inline void put(Writer &w, const Material &m)
{
    switch (m.kind) {
    case Material::Emissive: w.variant("Emissive"); w.array(1); put(w, m.spectrum); break;
    case Material::DirectionalEmissive:
        w.variant("DirectionalEmissive"); w.array(3); put(w, m.spectrum); w.f64(m.angle); w.f64(m.d); break;
    case Material::Diffuse: w.variant("Diffuse"); w.array(1); put(w, m.spectrum); break;
    case Material::Reflective: w.str("Reflective"); break;
    case Material::Dielectric: w.variant("Dielectric"); w.array(2); put(w, m.ior_); put(w, m.spectrum); break;
    }
}
inline Material get_material(Reader &r)
{
    static const char *const names[] = { "Emissive", "DirectionalEmissive", "Diffuse", "Reflective", "Dielectric" };
    bool payload;
    const int v = get_variant(r, names, 5, "Material", payload);
    Material m;
    m.kind = (Material::Kind)v;
    if (v == Material::Reflective) {
        if (payload) r.skip(); // a unit variant carries nothing; tolerate an empty payload
        return m;
    }
    if (!payload) r.fail("Material: variant without data");
    switch (v) {
    case Material::Emissive: r.expect_array(1, "Emissive"); m.spectrum = get_spectrum(r); break;
    case Material::DirectionalEmissive:
        r.expect_array(3, "DirectionalEmissive");
        m.spectrum = get_spectrum(r); m.angle = r.read_float(); m.d = r.read_float();
        break;
    case Material::Diffuse: r.expect_array(1, "Diffuse"); m.spectrum = get_spectrum(r); break;
    default: r.expect_array(2, "Dielectric"); m.ior_ = get_ior(r); m.spectrum = get_spectrum(r); break;
    }
    return m;
}

inline void put(Writer &w, const World &world)
{
    w.array(3);
    w.array(2); put(w, world.camera.center); put(w, world.camera.size);
    w.array(world.objects.size());
    for (const Object &o : world.objects) { w.array(2); put(w, o.shape); put(w, o.mat); }
    // the empty root World::new creates (rt_common.rs:550-553)
    w.array(6);
    w.array(0);
    w.array(2); put(w, DVec2(800.0 / 2.0, 600.0 / 2.0)); put(w, DVec2(800.0 / 2.0, 600.0 / 2.0));
    for (int k = 0; k < 4; ++k) w.nil();
}
inline World get_world(Reader &r)
{
    r.expect_array(3, "World");
    World world;
    r.expect_array(2, "Camera");
    world.camera.center = get_dvec2(r); world.camera.size = get_dvec2(r);
    const size_t n = r.read_array_len();
    world.objects.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        r.expect_array(2, "Object");
        Shape s = get_shape(r);
        Material m = get_material(r);
        world.objects.push_back(Object{ std::move(s), m });
    }
    r.skip(); // quadtree
    return world;
}

inline void put(Writer &w, const RawImage &img)
{
    w.array(3);
    w.sint(img.width); w.sint(img.height);
    w.array(img.data.size());
    // a PixelData is always the same 22 bytes: [weight, [x, y, z]] with four float 32
    w.out.reserve(w.out.size() + img.data.size() * 22);
    for (const PixelData &p : img.data) {
        uint8_t rec[22] = { 0x92, 0xca, 0, 0, 0, 0, 0x93, 0xca, 0, 0, 0, 0, 0xca, 0, 0, 0, 0, 0xca, 0, 0, 0, 0 };
        const float v[4] = { p.weight, p.value.x, p.value.y, p.value.z };
        const int at[4] = { 2, 8, 13, 18 };
        for (int k = 0; k < 4; ++k) {
            uint32_t u;
            memcpy(&u, &v[k], 4);
            rec[at[k]] = (uint8_t)(u >> 24); rec[at[k] + 1] = (uint8_t)(u >> 16); rec[at[k] + 2] = (uint8_t)(u >> 8); rec[at[k] + 3] = (uint8_t)u;
        }
        w.out.insert(w.out.end(), rec, rec + 22);
    }
}
inline RawImage get_raw_image(Reader &r)
{
    r.expect_array(3, "RawImage");
    const int64_t w = r.read_int(), h = r.read_int();
    if (w < 0 || h < 0 || w > INT32_MAX || h > INT32_MAX) r.fail("RawImage: bad size");
    const size_t n = r.read_array_len();
    if (n != (size_t)w * (size_t)h) r.fail("RawImage: data length is not width x height");
    RawImage img((int32_t)w, (int32_t)h);
    for (size_t i = 0; i < n; ++i) {
        r.expect_array(2, "PixelData");
        img.data[i].weight = (float)r.read_float();
        r.expect_array(3, "Vec3");
        img.data[i].value.x = (float)r.read_float(); img.data[i].value.y = (float)r.read_float(); img.data[i].value.z = (float)r.read_float();
    }
    return img;
}

// gui.rs:24-28
struct CheckPoint {
    World world;
    RawImage raw_image{ 0, 0 };
};

inline std::vector<uint8_t> encode_checkpoint(const CheckPoint &c)
{
    Writer w;
    w.array(2);
    put(w, c.world);
    put(w, c.raw_image);
    return std::move(w.out);
}
inline CheckPoint decode_checkpoint(const uint8_t *data, size_t size)
{
    Reader r(data, size);
    r.expect_array(2, "CheckPoint");
    CheckPoint c;
    c.world = get_world(r);
    c.raw_image = get_raw_image(r);
    if (!r.at_end()) r.fail("trailing bytes");
    return c;
}
inline std::vector<uint8_t> encode_world(const World &world) { Writer w; put(w, world); return std::move(w.out); }
inline World decode_world(const uint8_t *data, size_t size)
{
    Reader r(data, size);
    World world = get_world(r);
    if (!r.at_end()) r.fail("trailing bytes");
    return world;
}

// gui.rs:31-45
inline void save_checkpoint(const CheckPoint &c, const std::string &path)
{
    const std::vector<uint8_t> bytes = encode_checkpoint(c);
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path);
    const size_t n = fwrite(bytes.data(), 1, bytes.size(), f);
    if (fclose(f) != 0 || n != bytes.size()) throw std::runtime_error("cannot write " + path);
}
inline CheckPoint load_checkpoint(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> bytes;
    uint8_t buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    fclose(f);
    return decode_checkpoint(bytes.data(), bytes.size());
}

} // namespace checkpoint
} // namespace rt2d_host
