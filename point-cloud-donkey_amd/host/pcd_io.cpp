// pcd_io.cpp — point cloud file loading for ImplicitShapeModel::loadPointCloud (reference: implicit_shape_model.cpp:213-249
// loads .pcd / .ply into PointXYZRGBNormal through PCL). Built here: PCD v0.7 "ascii" and "binary" (not binary_compressed),
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
}  // namespace

std::shared_ptr<PointCloud> ImplicitShapeModel::loadPointCloud(const std::string& file) {
    const std::string ext = file.size() > 4 ? file.substr(file.size() - 4) : "";
    if (ext != ".pcd") { std::cerr << "ERROR: unknown or unsupported point cloud format (built: .pcd): " << file << std::endl; return nullptr; }
    std::ifstream in(file, std::ios::binary);
    if (!in) { std::cerr << "ERROR: could not load point cloud: " << file << std::endl; return nullptr; }
    std::vector<Field> fields;
    size_t points = 0; std::string data;
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
    } else { std::cerr << "ERROR: PCD DATA \"" << data << "\" is not built (ascii, binary): " << file << std::endl; return nullptr; }
    return cloud;
}

}  // namespace ism3d
