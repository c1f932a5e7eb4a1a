// config.cpp -- TOML scene loader, counterpart of ConfigModel (config.rs).
//
// The reference parses with the `toml` crate; this file carries a parser for the subset of
// TOML v1 scene files use: comments, bare/quoted/dotted keys, [tables], [[arrays of tables]],
// inline tables, (nested, multi-line) arrays, basic and literal strings, booleans, integers
// (decimal, 0x, 0o, 0b, with _) and floats.  Dates and multi-line strings are rejected.
//
// Beyond config.rs: the `Image` texture, and `Disc` / `Triangle` objects are built here (the
// reference parses them and then hits todo!() at config.rs:385, :617-622; `Triangle` has no
// parse arm at all).
//
// load_scene_file() reads a superset of that format which can describe a whole render, so that
// a workload is a data file: a [camera] table with the CameraBuilder fields (camera.rs:134-171),
// colours as [r, g, b] floats, `noise` textures with explicit tables, checkerboards over named
// textures, and for objects: `Box` (box3, hittable.rs:398-455), `ConstantMedium` with an inline
// boundary, `Group` (a HittableVec, or a BVHNode with bvh = true) with inline children,
// `center2` on a sphere (moving centre), `rotate_y` / `translate` on any object (applied in that
// order, as main.rs:45-48 does) and `light = true` to put an object into the `lights` argument
// of Camera::render.
#include "raytracing.hpp"

#include <cctype>
#include <cmath>
#include <cstdlib>
#include <sstream>

namespace raytracing {
namespace {

struct Value;
using Table = std::map<std::string, Value>; // toml::Table is a BTreeMap: iteration in key order
struct Value {
    enum Type { STRING, INTEGER, FLOAT, BOOLEAN, ARRAY, TABLE } type = TABLE;
    std::string s;
    int64_t i = 0;
    double f = 0;
    bool b = false;
    std::vector<Value> arr;
    std::shared_ptr<Table> tab;
    bool defined_inline = false;   // inline tables and arrays of values are closed for extension
    bool array_of_tables = false;
};

struct Parser {
    const std::string &src;
    size_t p = 0;
    int line = 1;
    explicit Parser(const std::string &s) : src(s) {}

    [[noreturn]] void err(const std::string &m) { throw Error("TOML parse error at line " + std::to_string(line) + ": " + m); }
    bool eof() const { return p >= src.size(); }
    char peek() const { return eof() ? '\0' : src[p]; }
    void skip_ws() {
        while (!eof() && (src[p] == ' ' || src[p] == '\t')) ++p;
    }
    void skip_comment() {
        if (peek() == '#')
            while (!eof() && src[p] != '\n') ++p;
    }
    void skip_ws_nl() { // inside arrays
        for (;;) {
            skip_ws();
            skip_comment();
            if (peek() == '\r') ++p;
            if (peek() == '\n') {
                ++p;
                ++line;
            } else
                break;
        }
    }
    void end_of_line() {
        skip_ws();
        skip_comment();
        if (peek() == '\r') ++p;
        if (eof()) return;
        if (peek() != '\n') err("expected end of line");
        ++p;
        ++line;
    }

    std::string basic_string() {
        ++p; // opening quote
        if (src.compare(p, 2, "\"\"") == 0) err("multi-line strings are not supported");
        std::string out;
        for (;;) {
            if (eof() || peek() == '\n') err("unterminated string");
            char c = src[p++];
            if (c == '"') break;
            if (c == '\\') {
                if (eof()) err("unterminated escape");
                char e = src[p++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case '"': out += '"'; break;
                case '\\': out += '\\'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                default: err("unsupported escape sequence");
                }
            } else
                out += c;
        }
        return out;
    }
    std::string literal_string() {
        ++p;
        if (src.compare(p, 2, "''") == 0) err("multi-line strings are not supported");
        std::string out;
        for (;;) {
            if (eof() || peek() == '\n') err("unterminated string");
            char c = src[p++];
            if (c == '\'') break;
            out += c;
        }
        return out;
    }

    std::string key_part() {
        skip_ws();
        if (peek() == '"') return basic_string();
        if (peek() == '\'') return literal_string();
        std::string k;
        while (!eof() && (std::isalnum((unsigned char)peek()) || peek() == '_' || peek() == '-')) k += src[p++];
        if (k.empty()) err("expected a key");
        return k;
    }
    std::vector<std::string> dotted_key() {
        std::vector<std::string> parts;
        parts.push_back(key_part());
        skip_ws();
        while (peek() == '.') {
            ++p;
            parts.push_back(key_part());
            skip_ws();
        }
        return parts;
    }

    Value number_or_date() {
        size_t start = p;
        while (!eof() && (std::isalnum((unsigned char)peek()) || peek() == '+' || peek() == '-' || peek() == '.' || peek() == '_' ||
                          peek() == ':'))
            ++p;
        std::string tok = src.substr(start, p - start);
        if (tok.empty()) err("expected a value");
        std::string t;
        for (size_t k = 0; k < tok.size(); ++k) {
            if (tok[k] == '_') {
                if (k == 0 || k + 1 == tok.size() || !std::isalnum((unsigned char)tok[k - 1]) || !std::isalnum((unsigned char)tok[k + 1]))
                    err("misplaced underscore in number");
                continue;
            }
            t += tok[k];
        }
        Value v;
        auto special = [&](const std::string &s) {
            std::string u = s;
            if (!u.empty() && (u[0] == '+' || u[0] == '-')) u = u.substr(1);
            return u == "inf" || u == "nan";
        };
        if (special(t)) {
            v.type = Value::FLOAT;
            bool neg = t[0] == '-';
            std::string u = (t[0] == '+' || t[0] == '-') ? t.substr(1) : t;
            v.f = u == "inf" ? (neg ? -HUGE_VAL : HUGE_VAL) : std::nan("");
            return v;
        }
        auto radix = [&](int base, const std::string &digits) {
            if (digits.empty()) err("empty integer literal");
            char *end = nullptr;
            errno = 0;
            long long x = std::strtoll(digits.c_str(), &end, base);
            if (*end || errno) err("bad integer literal '" + tok + "'");
            v.type = Value::INTEGER;
            v.i = x;
        };
        if (t.size() > 2 && t[0] == '0' && (t[1] == 'x' || t[1] == 'o' || t[1] == 'b')) {
            radix(t[1] == 'x' ? 16 : (t[1] == 'o' ? 8 : 2), t.substr(2));
            return v;
        }
        bool is_float = t.find_first_of(".eE") != std::string::npos;
        if (t.find(':') != std::string::npos || (t.size() >= 10 && t[4] == '-' && t[7] == '-')) err("dates and times are not supported");
        size_t digits_from = (t[0] == '+' || t[0] == '-') ? 1 : 0;
        if (digits_from >= t.size() || !std::isdigit((unsigned char)t[digits_from])) err("bad value '" + tok + "'");
        if (!is_float && t.size() > digits_from + 1 && t[digits_from] == '0') err("leading zeros are not allowed");
        char *end = nullptr;
        errno = 0;
        if (is_float) {
            v.type = Value::FLOAT;
            v.f = std::strtod(t.c_str(), &end);
        } else {
            v.type = Value::INTEGER;
            v.i = std::strtoll(t.c_str(), &end, 10);
        }
        if (*end || errno) err("bad number '" + tok + "'");
        return v;
    }

    Value value() {
        skip_ws();
        Value v;
        char c = peek();
        if (c == '"') {
            v.type = Value::STRING;
            v.s = basic_string();
        } else if (c == '\'') {
            v.type = Value::STRING;
            v.s = literal_string();
        } else if (c == '[') {
            ++p;
            v.type = Value::ARRAY;
            v.defined_inline = true;
            for (;;) {
                skip_ws_nl();
                if (peek() == ']') {
                    ++p;
                    break;
                }
                v.arr.push_back(value());
                skip_ws_nl();
                if (peek() == ',') {
                    ++p;
                    continue;
                }
                if (peek() == ']') {
                    ++p;
                    break;
                }
                err("expected ',' or ']' in array");
            }
        } else if (c == '{') {
            ++p;
            v.type = Value::TABLE;
            v.tab = std::make_shared<Table>();
            v.defined_inline = true;
            skip_ws();
            if (peek() == '}') {
                ++p;
                return v;
            }
            for (;;) {
                auto key = dotted_key();
                skip_ws();
                if (peek() != '=') err("expected '=' in inline table");
                ++p;
                Value item = value();
                insert(*v.tab, key, std::move(item));
                skip_ws();
                if (peek() == ',') {
                    ++p;
                    continue;
                }
                if (peek() == '}') {
                    ++p;
                    break;
                }
                err("expected ',' or '}' in inline table");
            }
        } else if (src.compare(p, 4, "true") == 0 && !std::isalnum((unsigned char)(p + 4 < src.size() ? src[p + 4] : ' '))) {
            p += 4;
            v.type = Value::BOOLEAN;
            v.b = true;
        } else if (src.compare(p, 5, "false") == 0 && !std::isalnum((unsigned char)(p + 5 < src.size() ? src[p + 5] : ' '))) {
            p += 5;
            v.type = Value::BOOLEAN;
            v.b = false;
        } else {
            v = number_or_date();
        }
        return v;
    }

    // walk/create intermediate tables for a dotted key
    Table &descend(Table &root, const std::vector<std::string> &path, size_t upto) {
        Table *t = &root;
        for (size_t k = 0; k < upto; ++k) {
            auto it = t->find(path[k]);
            if (it == t->end()) {
                Value nv;
                nv.type = Value::TABLE;
                nv.tab = std::make_shared<Table>();
                it = t->emplace(path[k], std::move(nv)).first;
            }
            Value &v = it->second;
            if (v.type == Value::ARRAY && v.array_of_tables) {
                t = v.arr.back().tab.get();
            } else if (v.type == Value::TABLE && !v.defined_inline) {
                t = v.tab.get();
            } else {
                err("key '" + path[k] + "' is not a table");
            }
        }
        return *t;
    }
    void insert(Table &root, const std::vector<std::string> &key, Value v) {
        Table &t = descend(root, key, key.size() - 1);
        if (t.count(key.back())) err("duplicate key '" + key.back() + "'");
        t.emplace(key.back(), std::move(v));
    }

    Table parse() {
        Table root;
        Table *cur = &root;
        while (!eof()) {
            skip_ws();
            char c = peek();
            if (c == '\n' || c == '\r' || c == '#' || c == '\0') {
                end_of_line();
                continue;
            }
            if (c == '[') {
                bool aot = src.compare(p, 2, "[[") == 0;
                p += aot ? 2 : 1;
                auto key = dotted_key();
                skip_ws();
                if (src.compare(p, aot ? 2 : 1, aot ? "]]" : "]") != 0) err("unterminated table header");
                p += aot ? 2 : 1;
                Table &parent = descend(root, key, key.size() - 1);
                auto it = parent.find(key.back());
                if (aot) {
                    if (it == parent.end()) {
                        Value nv;
                        nv.type = Value::ARRAY;
                        nv.array_of_tables = true;
                        it = parent.emplace(key.back(), std::move(nv)).first;
                    } else if (!(it->second.type == Value::ARRAY && it->second.array_of_tables)) {
                        err("'" + key.back() + "' is not an array of tables");
                    }
                    Value elem;
                    elem.type = Value::TABLE;
                    elem.tab = std::make_shared<Table>();
                    it->second.arr.push_back(std::move(elem));
                    cur = it->second.arr.back().tab.get();
                } else {
                    if (it == parent.end()) {
                        Value nv;
                        nv.type = Value::TABLE;
                        nv.tab = std::make_shared<Table>();
                        it = parent.emplace(key.back(), std::move(nv)).first;
                    } else if (it->second.type != Value::TABLE || it->second.defined_inline) {
                        err("cannot redefine '" + key.back() + "' as a table");
                    }
                    cur = it->second.tab.get();
                }
                end_of_line();
                continue;
            }
            auto key = dotted_key();
            skip_ws();
            if (peek() != '=') err("expected '=' after key");
            ++p;
            Value v = value();
            insert(*cur, key, std::move(v));
            end_of_line();
        }
        return root;
    }
};

// ------------------------------------------------------------------ ValueExt (config.rs:119-269)
std::string upper(std::string s) {
    for (auto &c : s) c = char(std::toupper((unsigned char)c));
    return s;
}

Color parse_color(const Value &v, const std::string &key) { // config.rs:132-145
    if (v.type == Value::STRING) {
        std::string s = v.s;
        size_t a = s.find_first_not_of(" \t\n\r"), b = s.find_last_not_of(" \t\n\r");
        s = a == std::string::npos ? "" : s.substr(a, b - a + 1);
        while (!s.empty() && s.front() == '#') s.erase(s.begin());
        while (!s.empty() && s.back() == '#') s.pop_back();
        bool ok = !s.empty() && s.size() <= 8;
        size_t start = (!s.empty() && s[0] == '+') ? 1 : 0; // from_str_radix accepts a leading '+'
        for (size_t k = start; k < s.size(); ++k) ok = ok && std::isxdigit((unsigned char)s[k]);
        if (!ok || s.size() == start) throw Error("Invalid hex string \"" + s + "\"");
        return Color::hex(uint32_t(std::strtoul(s.c_str() + start, nullptr, 16)));
    }
    if (v.type == Value::INTEGER) {
        if (v.i < 0 || v.i > 0xFFFFFFFFll) throw Error(key + ": color integer out of range"); // try_into().unwrap() panics
        return Color::hex(uint32_t(v.i));
    }
    throw Error(key + " must be a hex code or number.");
}
// scene files also take [r, g, b] floats: emission 20 * white or random albedos are not 8-bit
Color parse_color_ext(const Value &v, const std::string &key) {
    if (v.type == Value::ARRAY && !v.array_of_tables) {
        if (v.arr.size() != 3) throw Error(key + " must be a hex code, a number or [r, g, b].");
        double c[3];
        for (int i = 0; i < 3; ++i) {
            if (v.arr[i].type == Value::FLOAT) c[i] = v.arr[i].f;
            else if (v.arr[i].type == Value::INTEGER) c[i] = double(v.arr[i].i);
            else throw Error(key + "." + std::to_string(i) + " must be a decimal number.");
        }
        return Color{c[0], c[1], c[2]};
    }
    return parse_color(v, key);
}
double parse_floatlike(const Value &v, const std::string &key) { // config.rs:147-156
    if (v.type == Value::FLOAT) return v.f;
    if (v.type == Value::INTEGER) return double(v.i);
    throw Error(key + " must be a decimal number.");
}
Vec3 parse_vec3(const Value &v, const std::string &key, const char *what) { // config.rs:167-217
    if (v.type != Value::ARRAY || v.array_of_tables)
        throw Error(key + " must be a valid 3D " + what + ", represented as [x, y, z].");
    if (v.arr.size() != 3) throw Error(key + " must be a valid 3D " + what + ", represented as [x, y, z].");
    double r[3];
    for (int i = 0; i < 3; ++i) r[i] = parse_floatlike(v.arr[i], key + "." + std::to_string(i));
    return {r[0], r[1], r[2]};
}
const std::vector<Value> &parse_array(const Value &v, const std::string &key) { // config.rs:219-226
    if (v.type != Value::ARRAY) throw Error(key + " must be an array.");
    return v.arr;
}
const Value &require_value(const Table &t, const std::string &key, const std::string &parent) { // config.rs:271-284
    auto it = t.find(key);
    if (it == t.end()) throw Error(parent + "." + key + " must be provided.");
    return it->second;
}
const std::string *get_type(const Table &t) {
    auto it = t.find("type");
    if (it == t.end() || it->second.type != Value::STRING) return nullptr;
    return &it->second.s;
}

struct Loader {
    std::map<std::string, TexturePtr> textures;
    std::map<std::string, MaterialPtr> materials;
    bool extended = false;      // scene-file superset
    std::string base_dir;       // image paths of a scene file are relative to it
    int *next_prim_id = nullptr; // primitives are numbered in the order they are created
    std::vector<HittablePtr> lights;

    Color color(const Value &v, const std::string &key) const { return extended ? parse_color_ext(v, key) : parse_color(v, key); }
    template <typename T> std::shared_ptr<T> numbered(std::shared_ptr<T> h) {
        if (next_prim_id) h->prim_id = (*next_prim_id)++;
        return h;
    }
    // a material given by name, or (scene files) a texture given by name or an inline colour
    TexturePtr texture_or_color(const Table &t, const std::string &base) {
        auto it = t.find("texture");
        if (it != t.end()) return texture_by_name(it->second, base + ".texture");
        return std::make_shared<SolidColor>(color(require_value(t, "color", base), base + ".color"));
    }

    TexturePtr texture_by_name(const Value &v, const std::string &key) { // config.rs:228-247
        if (v.type != Value::STRING) throw Error(key + " must be a string representing a previously listed texture.");
        auto it = textures.find(v.s);
        if (it == textures.end())
            throw Error(key + " does not describe a valid texture. (No texture with ID " + v.s + " has been loaded.)");
        return it->second;
    }
    MaterialPtr material_by_name(const Value &v, const std::string &key) { // config.rs:249-268
        if (v.type != Value::STRING) throw Error(key + " must be a string representing a previously listed texture.");
        auto it = materials.find(v.s);
        if (it == materials.end())
            throw Error(key + " does not describe a valid texture. (No material named " + v.s + " has been loaded.)");
        return it->second;
    }

    TexturePtr texture(const std::string &name, const Table &t) { // config.rs:286-388
        const std::string *ty = get_type(t);
        std::string base = "config.textures." + name;
        if (!ty) throw Error(base + ".type must be a string.");
        std::string T = upper(*ty);
        if (T == "COLOR" || T == "SOLIDCOLOR" || T == "SOLID_COLOR") {
            return std::make_shared<SolidColor>(color(require_value(t, "color", base), base + ".color"));
        }
        if (T == "CHECKERBOARD" || T == "CHECKER") {
            double scale = parse_floatlike(require_value(t, "scale", base), base + ".scale");
            const auto &arr = parse_array(require_value(t, "textures", base), base + ".textures");
            if (arr.size() != 2) throw Error(base + ".textures must be an array of length 2.");
            if (extended && (arr[0].type == Value::STRING || arr[1].type == Value::STRING)) {
                // names of textures listed earlier in the file (the reference takes two colours)
                TexturePtr sub[2];
                for (int k = 0; k < 2; ++k) {
                    const std::string key = base + ".textures." + std::to_string(k);
                    auto named = arr[k].type == Value::STRING ? textures.find(arr[k].s) : textures.end();
                    sub[k] = named != textures.end() ? named->second : std::make_shared<SolidColor>(color(arr[k], key));
                }
                return std::make_shared<Checkerboard>(scale, sub[0], sub[1]);
            }
            Color c0 = color(arr[0], base + ".textures.0");
            Color c1 = color(arr[1], base + ".textures.1");
            return Checkerboard::solid(scale, c0, c1);
        }
        if (T == "IMAGE") {
            const Value &pv = require_value(t, "path", base);
            if (pv.type != Value::STRING) throw Error(base + ".path must be a valid filepath.");
            const bool relative = !pv.s.empty() && pv.s[0] != '/';
            // throws "Failed to find file ..." like config.rs:347-356
            return ImageTexture::load(extended && relative && !base_dir.empty() ? base_dir + "/" + pv.s : pv.s);
        }
        if (extended && (T == "NOISE" || T == "PERLIN")) {
            double scale = parse_floatlike(require_value(t, "scale", base), base + ".scale");
            const auto &rv = parse_array(require_value(t, "ranvec", base), base + ".ranvec");
            const auto &pm = parse_array(require_value(t, "perm", base), base + ".perm");
            if (rv.size() != 256 || pm.size() != 3) throw Error(base + ": ranvec must hold 256 vectors and perm 3 permutations.");
            std::vector<double> vec(768);
            std::vector<int32_t> perm(768);
            for (int i = 0; i < 256; ++i) {
                Vec3 v = parse_vec3(rv[i], base + ".ranvec." + std::to_string(i), "vector");
                vec[3 * i] = v.x, vec[3 * i + 1] = v.y, vec[3 * i + 2] = v.z;
            }
            for (int a = 0; a < 3; ++a) {
                const auto &row = parse_array(pm[a], base + ".perm." + std::to_string(a));
                if (row.size() != 256) throw Error(base + ".perm rows must hold 256 entries.");
                for (int i = 0; i < 256; ++i) {
                    if (row[i].type != Value::INTEGER || row[i].i < 0 || row[i].i > 255)
                        throw Error(base + ".perm entries must be integers 0..255.");
                    perm[a * 256 + i] = int32_t(row[i].i);
                }
            }
            return std::make_shared<NoiseTexture>(scale, vec.data(), perm.data());
        }
        throw Error(base + ".type must be a valid texture type. (valid: \"color\" | \"checkerboard\" | \"image\")");
    }

    MaterialPtr material(const std::string &name, const Table &t) { // config.rs:390-496
        const std::string *ty = get_type(t);
        std::string base = "config.materials." + name;
        if (!ty) throw Error(base + ".type must be a string.");
        std::string T = upper(*ty);
        if (T == "LAMBERTIAN")
            return std::make_shared<Lambertian>(extended ? texture_or_color(t, base)
                                                         : texture_by_name(require_value(t, "texture", base), base + ".texture"));
        if (T == "METAL" || T == "METALLIC" || T == "FUZZY") {
            Color albedo = color(require_value(t, "albedo", base), base + ".albedo");
            double fuzz = parse_floatlike(require_value(t, "fuzz", base), base + ".fuzz");
            if (fuzz > 1.0) fuzz /= 100.0; // config.rs:417-419
            return Metal::with_fuzz(albedo, fuzz);
        }
        if (T == "LIGHT" || T == "LIGHTSOURCE" || T == "DIFFUSELIGHT")
            return std::make_shared<DiffuseLight>(extended ? texture_or_color(t, base)
                                                           : texture_by_name(require_value(t, "texture", base), base + ".texture"));
        if (T == "DIELECTRIC")
            return std::make_shared<Dielectric>(parse_floatlike(require_value(t, "refractive_index", base), base + ".refractive_index"));
        if (T == "ISOTROPIC")
            return std::make_shared<Isotropic>(extended ? texture_or_color(t, base)
                                                        : texture_by_name(require_value(t, "texture", base), base + ".texture"));
        if (T == "SOLIDCOLOR") return Lambertian::solid(color(require_value(t, "color", base), base + ".color"));
        if (T == "COLOREDLIGHT") {
            Color c = color(require_value(t, "color", base), base + ".color");
            auto it = t.find("brightness");
            if (it != t.end()) c.set_brightness(parse_floatlike(it->second, base + ".brightness"));
            return DiffuseLight::solid(c);
        }
        throw Error(base + ".type must be a valid material type. (valid: \"metal\" | \"light\" | \"lambertian\" | \"dielectric\")");
    }

    std::map<std::string, HittablePtr> named_objects; // scene files: objects with a `name`

    HittablePtr object(size_t index, const Table &t) { // config.rs:498-625
        const std::string *ty = get_type(t);
        if (!ty) throw Error("config.materials." + std::to_string(index) + ".type must be a string."); // sic, config.rs:508
        return object_at("config.objects." + std::to_string(index), t);
    }

    // an inline object table, or (scene files) the name of an object listed earlier
    HittablePtr object_value(const Value &v, const std::string &base) {
        if (v.type == Value::STRING) {
            auto it = named_objects.find(v.s);
            if (it == named_objects.end()) throw Error(base + " names an object that has not been listed (" + v.s + ").");
            return it->second;
        }
        if (v.type != Value::TABLE) throw Error(base + " must be a table or the name of an object.");
        return object_at(base, *v.tab);
    }

    HittablePtr object_at(const std::string &base, const Table &t) {
        const std::string *ty = get_type(t);
        if (!ty) throw Error(base + ".type must be a string.");
        HittablePtr o = bare_object(base, upper(*ty), t);
        if (!extended) return o;
        auto it = t.find("rotate_y"); // instance transforms, in the order of main.rs:45-48
        if (it != t.end()) // degrees, converted as f64::to_radians does (main.rs:45)
            o = std::make_shared<RotateY>(o, parse_floatlike(it->second, base + ".rotate_y") * (3.14159265358979323846 / 180.0));
        it = t.find("translate");
        if (it != t.end()) o = std::make_shared<Translate>(o, parse_vec3(it->second, base + ".translate", "vector"));
        it = t.find("light");
        if (it != t.end()) {
            if (it->second.type != Value::BOOLEAN) throw Error(base + ".light must be true or false.");
            if (it->second.b) lights.push_back(o);
        }
        it = t.find("name");
        if (it != t.end()) {
            if (it->second.type != Value::STRING) throw Error(base + ".name must be a string.");
            named_objects[it->second.s] = o;
        }
        return o;
    }

    HittablePtr bare_object(const std::string &base, const std::string &T, const Table &t) {
        auto two_vectors = [&](Vec3 out[2]) {
            const auto &vecs = parse_array(require_value(t, "vectors", base), base + ".vectors");
            if (vecs.size() != 2) throw Error(base + ".vectors must be an array of length 2.");
            out[0] = parse_vec3(vecs[0], base + ".vectors.0", "vector");
            out[1] = parse_vec3(vecs[1], base + ".vectors.1", "vector");
        };
        if (T == "SPHERE") {
            Point3 c = parse_vec3(require_value(t, "center", base), base + ".center", "point");
            double r = parse_floatlike(require_value(t, "radius", base), base + ".radius");
            MaterialPtr m = material_by_name(require_value(t, "material", base), base + ".material");
            auto cv = t.find("center_vec"); // scene files: centre(time) = center + time * center_vec (hittable.rs:130-148)
            if (extended && cv != t.end())
                return numbered(std::make_shared<Sphere>(c, parse_vec3(cv->second, base + ".center_vec", "vector"), r, m));
            return numbered(Sphere::stationary(c, r, m));
        }
        if (T == "PARALLELOGRAM") {
            Point3 c = parse_vec3(require_value(t, "corner", base), base + ".corner", "point");
            MaterialPtr m = material_by_name(require_value(t, "material", base), base + ".material");
            Vec3 v[2];
            two_vectors(v);
            return numbered(std::make_shared<Parallelogram>(c, v[0], v[1], m));
        }
        if (T == "DISC") {
            Point3 c = parse_vec3(require_value(t, "center", base), base + ".center", "point");
            MaterialPtr m = material_by_name(require_value(t, "material", base), base + ".material");
            Vec3 v[2];
            two_vectors(v);
            return numbered(DiscFromCenter(c, v[0], v[1], m));
        }
        if (T == "TRIANGLE") {
            const auto &pts = parse_array(require_value(t, "points", base), base + ".points");
            if (pts.size() != 3) throw Error(base + ".points must be an array of length 3.");
            MaterialPtr m = material_by_name(require_value(t, "material", base), base + ".material");
            Point3 p[3];
            for (int i = 0; i < 3; ++i) p[i] = parse_vec3(pts[i], base + ".points." + std::to_string(i), "point");
            return numbered(TriangleFromPoints(p[0], p[1], p[2], m));
        }
        if (extended && T == "BOX") { // box3 (hittable.rs:398-455)
            Point3 a = parse_vec3(require_value(t, "min", base), base + ".min", "point");
            Point3 b = parse_vec3(require_value(t, "max", base), base + ".max", "point");
            MaterialPtr m = material_by_name(require_value(t, "material", base), base + ".material");
            auto box = box3(a, b, m);
            for (auto &side : box->objects) numbered(side);
            return box;
        }
        if (extended && (T == "CONSTANTMEDIUM" || T == "MEDIUM")) { // hittable.rs:820-845
            HittablePtr boundary = object_value(require_value(t, "boundary", base), base + ".boundary");
            double density = parse_floatlike(require_value(t, "density", base), base + ".density");
            return numbered(std::make_shared<ConstantMedium>(boundary, density, texture_or_color(t, base)));
        }
        if (extended && T == "GROUP") {
            const auto &kids = parse_array(require_value(t, "children", base), base + ".children");
            std::vector<HittablePtr> objs;
            for (size_t i = 0; i < kids.size(); ++i) objs.push_back(object_value(kids[i], base + ".children." + std::to_string(i)));
            bool bvh = false;
            auto it = t.find("bvh");
            if (it != t.end()) {
                if (it->second.type != Value::BOOLEAN) throw Error(base + ".bvh must be true or false.");
                bvh = it->second.b;
            }
            if (bvh) {
                if (objs.empty()) throw Error(base + ": a BVH needs at least one child.");
                return std::make_shared<BVHNode>(std::move(objs)); // BVHNode::new (boundingbox.rs:179-206)
            }
            auto list = std::make_shared<HittableVec>();
            for (auto &o : objs) list->add(o);
            return list;
        }
        if (extended)
            throw Error(base + ".type must be a valid object type. (valid: \"sphere\" | \"parallelogram\" | \"triangle\" | "
                               "\"disc\" | \"box\" | \"constantmedium\" | \"group\")");
        throw Error(base + ".type must be a valid object type. (valid: \"sphere\" | \"parallelogram\" | \"triangle\" | \"disc\")");
    }
};

} // namespace

ConfigModel ConfigModel::from_str(const std::string &text) { // config.rs:628-706
    Parser parser(text);
    Table root = parser.parse();
    auto get_table = [&](const char *k) -> const Table & {
        auto it = root.find(k);
        if (it == root.end() || it->second.type != Value::TABLE) throw Error(std::string("config.") + k + " must be a table.");
        return *it->second.tab;
    };
    const Table &tex_table = get_table("textures");
    const Table &mat_table = get_table("materials");
    auto oit = root.find("objects");
    if (oit == root.end() || oit->second.type != Value::ARRAY) throw Error("config.objects must be a list of tables.");

    Loader L;
    for (const auto &kv : tex_table) {
        if (kv.second.type != Value::TABLE) throw Error("config.textures." + kv.first + " must be a table.");
        L.textures[kv.first] = L.texture(kv.first, *kv.second.tab);
    }
    for (const auto &kv : mat_table) {
        if (kv.second.type != Value::TABLE) throw Error("config.materials." + kv.first + " must be a table.");
        L.materials[kv.first] = L.material(kv.first, *kv.second.tab);
    }
    ConfigModel cfg;
    size_t i = 0;
    for (const auto &ov : oit->second.arr) {
        if (ov.type != Value::TABLE) throw Error("config.objects." + std::to_string(i) + " must be a table.");
        cfg.objects.push_back(L.object(i, *ov.tab));
        ++i;
    }
    cfg.materials = L.materials;
    return cfg;
}

SceneFile load_scene_file(const std::string &text, const std::string &base_dir, int &next_prim_id) {
    Parser parser(text);
    Table root = parser.parse();
    Loader L;
    L.extended = true;
    L.base_dir = base_dir;
    L.next_prim_id = &next_prim_id;
    auto table_of = [&](const char *k) -> const Table * { // textures / materials may be left out
        auto it = root.find(k);
        if (it == root.end()) return nullptr;
        if (it->second.type != Value::TABLE) throw Error(std::string("config.") + k + " must be a table.");
        return it->second.tab.get();
    };
    // checkerboards may name other textures, so textures are built in file order where the parser
    // kept it; the map is in key order (BTreeMap), so a named sub-texture must sort before its user
    if (const Table *tt = table_of("textures"))
        for (const auto &kv : *tt) {
            if (kv.second.type != Value::TABLE) throw Error("config.textures." + kv.first + " must be a table.");
            L.textures[kv.first] = L.texture(kv.first, *kv.second.tab);
        }
    if (const Table *mt = table_of("materials"))
        for (const auto &kv : *mt) {
            if (kv.second.type != Value::TABLE) throw Error("config.materials." + kv.first + " must be a table.");
            L.materials[kv.first] = L.material(kv.first, *kv.second.tab);
        }
    auto oit = root.find("objects");
    if (oit == root.end() || oit->second.type != Value::ARRAY) throw Error("config.objects must be a list of tables.");
    SceneFile out;
    auto world = std::make_shared<HittableVec>();
    size_t i = 0;
    for (const auto &ov : oit->second.arr) {
        if (ov.type != Value::TABLE) throw Error("config.objects." + std::to_string(i) + " must be a table.");
        world->add(L.object_at("config.objects." + std::to_string(i), *ov.tab));
        ++i;
    }
    out.world = world;
    out.materials = L.materials;
    // `lights`: one object as it is, several as a HittableVec (which draws the pick, pdf.rs);
    // light_list = true forces the list form for a single light
    bool force_list = false;
    auto ll = root.find("light_list");
    if (ll != root.end()) {
        if (ll->second.type != Value::BOOLEAN) throw Error("config.light_list must be true or false.");
        force_list = ll->second.b;
    }
    if (L.lights.size() == 1 && !force_list) {
        out.lights = L.lights[0];
    } else if (!L.lights.empty()) {
        auto lv = std::make_shared<HittableVec>();
        for (auto &l : L.lights) lv->add(l);
        out.lights = lv;
    }
    rth_camera_defaults(&out.camera);
    if (const Table *ct = table_of("camera")) { // CameraBuilder setters (camera.rs:57-171)
        out.has_camera = true;
        rth_camera_builder &c = out.camera;
        const std::string base = "config.camera";
        auto num = [&](const char *k, double &dst) {
            auto it = ct->find(k);
            if (it != ct->end()) dst = parse_floatlike(it->second, base + "." + k);
        };
        auto uint = [&](const char *k, uint32_t &dst) {
            auto it = ct->find(k);
            if (it == ct->end()) return;
            if (it->second.type != Value::INTEGER || it->second.i < 0 || it->second.i > 0xFFFFFFFFll)
                throw Error(base + "." + k + " must be a non-negative integer.");
            dst = uint32_t(it->second.i);
        };
        auto vec = [&](const char *k, double dst[3], const char *what) {
            auto it = ct->find(k);
            if (it == ct->end()) return;
            Vec3 v = parse_vec3(it->second, base + "." + k, what);
            dst[0] = v.x, dst[1] = v.y, dst[2] = v.z;
        };
        uint("image_width", c.image_width);
        uint("image_height", c.image_height);
        auto ar = ct->find("aspect_ratio");
        if (ar != ct->end()) // with_aspect_ratio (camera.rs:78-92): height = round(width / ratio)
            rth_camera_with_aspect_ratio(&c, c.image_width, parse_floatlike(ar->second, base + ".aspect_ratio"));
        uint("samples_per_px", c.samples_per_px);
        uint("max_depth", c.max_depth);
        num("vfov", c.vfov_deg);
        vec("camera_center", c.center, "point");
        vec("camera_target", c.target, "point");
        vec("vup", c.vup, "vector");
        num("defocus_angle", c.defocus_angle_deg);
        num("focal_length", c.focal_length);
        auto aa = ct->find("antialiasing"); // AntialiasingType (camera.rs:204-211)
        if (aa != ct->end()) {
            const std::string v = aa->second.type == Value::STRING ? upper(aa->second.s) : "";
            if (v != "SQUARE" && v != "DISC") throw Error(base + ".antialiasing must be \"square\" or \"disc\".");
            c.antialiasing = v == "DISC" ? int32_t(RT_AA_DISC) : int32_t(RT_AA_SQUARE);
        }
        auto bg = ct->find("background");
        if (bg != ct->end()) {
            if (bg->second.type == Value::STRING && upper(bg->second.s) == "SKY") {
                c.background_sky = 1;
            } else {
                Color col = parse_color_ext(bg->second, base + ".background");
                c.background_sky = 0;
                c.background[0] = col.r, c.background[1] = col.g, c.background[2] = col.b;
            }
        }
    }
    return out;
}

std::shared_ptr<HittableVec> ConfigModel::as_world() const { // config.rs:691-697
    auto world = std::make_shared<HittableVec>();
    for (const auto &o : objects) world->add(o);
    return world;
}

} // namespace raytracing
