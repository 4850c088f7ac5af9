// pcd_io.cpp — point cloud file loading for ImplicitShapeModel::loadPointCloud (reference: implicit_shape_model.cpp:213-249
// loads .pcd / .ply into PointXYZRGBNormal through PCL). Built here: PCD v0.7 "ascii", "binary" and "binary_compressed" (LZF),
// fields x y z [rgb|rgba] [normal_x normal_y normal_z] in any order; other fields are skipped. NaN points are removed
// (pcl::removeNaNFromPointCloud, implicit_shape_model.cpp:608-611). IO is outside the hot path (host only).
#include <cmath>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "ism3d.h"

namespace ism3d {

namespace {
struct Field { std::string name; int size = 4; char type = 'F'; int count = 1; int offset = 0; };

double readScalar(const char* p, const Field& f) {
    switch (f.type) {
        case 'F': if (f.size == 4) { float v; std::memcpy(&v, p, 4); return v; } else { double v; std::memcpy(&v, p, 8); return v; }
        case 'U': if (f.size == 1) return *(const uint8_t*)p; if (f.size == 2) { uint16_t v; std::memcpy(&v, p, 2); return v; } { uint32_t v; std::memcpy(&v, p, 4); return v; }
        case 'I': if (f.size == 1) return *(const int8_t*)p; if (f.size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; } { int32_t v; std::memcpy(&v, p, 4); return v; }
    }
    return 0;
}
// LZF (Marc Lehmann's format, as written by pcl::lzfCompress): a control byte < 32 starts a literal run of ctrl+1 bytes; otherwise
// it is a back reference of length (ctrl >> 5) + 2 (7 -> one more length byte) at distance ((ctrl & 31) << 8 | next byte) + 1.
bool lzfDecompress(const unsigned char* in, size_t in_len, unsigned char* out, size_t out_len) {
    size_t ip = 0, op = 0;
    while (ip < in_len) {
        unsigned ctrl = in[ip++];
        if (ctrl < 32) {
            const size_t run = ctrl + 1;
            if (ip + run > in_len || op + run > out_len) return false;
            std::memcpy(out + op, in + ip, run);
            ip += run; op += run;
        } else {
            size_t len = ctrl >> 5;
            if (len == 7) { if (ip >= in_len) return false; len += in[ip++]; }
            if (ip >= in_len) return false;
            const size_t dist = ((size_t)(ctrl & 0x1f) << 8 | in[ip++]) + 1;
            len += 2;
            if (dist > op || op + len > out_len) return false;
            for (size_t i = 0; i < len; ++i, ++op) out[op] = out[op - dist];      // overlapping copies repeat the pattern
        }
    }
    return op == out_len;
}
}  // namespace

std::shared_ptr<PointCloud> ImplicitShapeModel::loadPointCloud(const std::string& file) {
    const std::string ext = file.size() > 4 ? file.substr(file.size() - 4) : "";
    if (ext != ".pcd") { std::cerr << "ERROR: unknown or unsupported point cloud format (built: .pcd): " << file << std::endl; return nullptr; }
    std::ifstream in(file, std::ios::binary);
    if (!in) { std::cerr << "ERROR: could not load point cloud: " << file << std::endl; return nullptr; }
    std::vector<Field> fields;
    size_t points = 0, height = 1; std::string data;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ls(line);
        std::string key; ls >> key;
        if (key == "FIELDS") { std::string n; while (ls >> n) { Field f; f.name = n; fields.push_back(f); } }
        else if (key == "SIZE") { for (auto& f : fields) ls >> f.size; }
        else if (key == "TYPE") { for (auto& f : fields) ls >> f.type; }
        else if (key == "COUNT") { for (auto& f : fields) ls >> f.count; }
        else if (key == "POINTS") ls >> points;
        else if (key == "WIDTH" && points == 0) { size_t w; ls >> w; points = w; }
        else if (key == "HEIGHT") ls >> height;
        else if (key == "DATA") { ls >> data; break; }
    }
    if (fields.empty() || data.empty()) { std::cerr << "ERROR: malformed PCD header: " << file << std::endl; return nullptr; }
    int stride = 0;
    for (auto& f : fields) { f.offset = stride; stride += f.size * f.count; }
    auto idx = [&](const char* n) { for (size_t i = 0; i < fields.size(); ++i) if (fields[i].name == n) return (int)i; return -1; };
    const int ix = idx("x"), iy = idx("y"), iz = idx("z"), inx = idx("normal_x"), iny = idx("normal_y"), inz = idx("normal_z");
    int irgb = idx("rgb"); if (irgb < 0) irgb = idx("rgba");
    if (ix < 0 || iy < 0 || iz < 0) { std::cerr << "ERROR: PCD without x y z: " << file << std::endl; return nullptr; }
    auto cloud = std::make_shared<PointCloud>();
    auto push = [&](const std::vector<double>& v, uint32_t rgb) {
        const float px = (float)v[0], py = (float)v[1], pz = (float)v[2];
        if (!std::isfinite(px) || !std::isfinite(py) || !std::isfinite(pz)) return;
        cloud->x.push_back(px); cloud->y.push_back(py); cloud->z.push_back(pz);
        cloud->nx.push_back((float)v[3]); cloud->ny.push_back((float)v[4]); cloud->nz.push_back((float)v[5]);
        if (irgb >= 0) cloud->rgba.push_back(rgb & 0x00ffffffu);
    };
    if (data == "ascii") {
        while (std::getline(in, line)) {
            if (line.empty()) continue;
            std::istringstream ls(line);
            std::vector<double> v(6, 0.0); uint32_t rgb = 0;
            for (size_t f = 0; f < fields.size(); ++f)
                for (int c = 0; c < fields[f].count; ++c) {
                    std::string tok; ls >> tok;
                    double val = 0;
                    if ((int)f == irgb && fields[f].type == 'F') { float fv = std::strtof(tok.c_str(), nullptr); std::memcpy(&rgb, &fv, 4); }
                    else if ((int)f == irgb) rgb = (uint32_t)std::strtoul(tok.c_str(), nullptr, 10);
                    else val = (tok == "nan" || tok == "NaN") ? NAN : std::strtod(tok.c_str(), nullptr);
                    if (c == 0) { if ((int)f == ix) v[0] = val; else if ((int)f == iy) v[1] = val; else if ((int)f == iz) v[2] = val;
                                  else if ((int)f == inx) v[3] = val; else if ((int)f == iny) v[4] = val; else if ((int)f == inz) v[5] = val; }
                }
            push(v, rgb);
        }
    } else if (data == "binary") {
        std::vector<char> buf((size_t)stride * points);
        in.read(buf.data(), buf.size());
        const size_t got = (size_t)in.gcount() / stride;
        for (size_t p = 0; p < got; ++p) {
            const char* rec = buf.data() + p * stride;
            std::vector<double> v(6, 0.0); uint32_t rgb = 0;
            v[0] = readScalar(rec + fields[ix].offset, fields[ix]); v[1] = readScalar(rec + fields[iy].offset, fields[iy]); v[2] = readScalar(rec + fields[iz].offset, fields[iz]);
            if (inx >= 0) v[3] = readScalar(rec + fields[inx].offset, fields[inx]);
            if (iny >= 0) v[4] = readScalar(rec + fields[iny].offset, fields[iny]);
            if (inz >= 0) v[5] = readScalar(rec + fields[inz].offset, fields[inz]);
            if (irgb >= 0) std::memcpy(&rgb, rec + fields[irgb].offset, 4);
            push(v, rgb);
        }
    } else if (data == "binary_compressed") {
        // pcl::PCDWriter::writeBinaryCompressed: uint32 compressed size, uint32 uncompressed size, then one LZF stream whose
        // plain form is field-major (all x, then all y, ...: each field's size*count bytes per point, point after point)
        uint32_t csize = 0, usize = 0;
        in.read((char*)&csize, 4); in.read((char*)&usize, 4);
        if (!in || usize != (uint64_t)stride * points) { std::cerr << "ERROR: malformed binary_compressed PCD: " << file << std::endl; return nullptr; }
        std::vector<unsigned char> cbuf(csize), ubuf(usize);
        in.read((char*)cbuf.data(), csize);
        if ((size_t)in.gcount() != csize || !lzfDecompress(cbuf.data(), csize, ubuf.data(), usize)) {
            std::cerr << "ERROR: corrupt LZF stream in PCD: " << file << std::endl; return nullptr;
        }
        std::vector<size_t> base(fields.size());
        size_t off = 0;
        for (size_t f = 0; f < fields.size(); ++f) { base[f] = off; off += (size_t)fields[f].size * fields[f].count * points; }
        auto at = [&](int f, size_t p) { return (const char*)ubuf.data() + base[f] + p * (size_t)fields[f].size * fields[f].count; };
        for (size_t p = 0; p < points; ++p) {
            std::vector<double> v(6, 0.0); uint32_t rgb = 0;
            v[0] = readScalar(at(ix, p), fields[ix]); v[1] = readScalar(at(iy, p), fields[iy]); v[2] = readScalar(at(iz, p), fields[iz]);
            if (inx >= 0) v[3] = readScalar(at(inx, p), fields[inx]);
            if (iny >= 0) v[4] = readScalar(at(iny, p), fields[iny]);
            if (inz >= 0) v[5] = readScalar(at(inz, p), fields[inz]);
            if (irgb >= 0) std::memcpy(&rgb, at(irgb, p), 4);
            push(v, rgb);
        }
    } else { std::cerr << "ERROR: PCD DATA \"" << data << "\" is not built (ascii, binary, binary_compressed): " << file << std::endl; return nullptr; }
    // pcl::removeNaNFromPointCloud keeps width x height only when nothing had to go: such a cloud stays "organized" in the reference
    cloud->organized = height > 1 && cloud->size() == points;
    return cloud;
}

}  // namespace ism3d
