// Host seam of the path tracer: scene load and camera, the code that sits in front of the
// ray-tracing pipeline in the reference (main.cpp:1101-1190 loadModel, main.cpp:1465-1480
// updateUniformBuffer).  Pure host C++, no GPU calls; exported through the C ABI of
// include/pt_hip.h.
//
// The OBJ/MTL reader is written from scratch but follows the observable behaviour of the
// tinyobjloader v1.0.6 header the reference vendors (external/tinyobjloader), because that
// behaviour defines primitive order, vertex order and the exact float bits of the scene:
//   * numbers: decimal digits accumulated in double, fraction digits weighted by a 1e-k table,
//     optional exponent applied as ldexp(m * 5^e, e), result narrowed to float
//     (tiny_obj_loader.h:474-597) -- deliberately NOT strtof;
//   * faces: i, i/j, i//k, i/j/k with 1-based or negative (relative) indices (:425-429,
//     :691-731), polygons triangulated as a fan (0,k-1,k) (:908-931);
//   * `usemtl` switching flushes the pending faces into the SAME shape (:1528-1555); `g`/`o`
//     start a new shape and drop a shape whose pending face list is empty (:1597-1660);
//   * a missing .mtl is only a warning (:1347-1355, :1587-1593).
// tests/test_host_seam.py pins all of this against the real library (tests/golden/ref_seam.npz).
#include "../../include/pt_hip.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <new>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

struct PtSceneData
{
	std::vector<PtVertex> vertices;
	std::vector<uint32_t> indices;
	std::vector<PtMaterial> materials;
	std::vector<uint32_t> matIndices;
	std::vector<std::string> diffuseMaps;  // per material: path of its map_Kd image (relative names resolved against the OBJ's directory), or empty
	std::string warnings;
};

static thread_local std::string g_seamError;
const char *pt_seam_last_error() { return g_seamError.c_str(); }

namespace
{
struct Cursor
{
	const char *p;
	const char *end;
	bool done() const { return p >= end; }
	void skipBlank()
	{
		while (p < end && (*p == ' ' || *p == '\t')) ++p;
	}
	// token = run of characters up to blank / CR
	void token(const char *&b, const char *&e)
	{
		skipBlank();
		b = p;
		while (p < end && *p != ' ' && *p != '\t' && *p != '\r') ++p;
		e = p;
	}
};

inline bool isDigit(char c) { return c >= '0' && c <= '9'; }

// Number grammar and arithmetic of tinyobjloader's parser (see file header).
bool parseNumber(const char *s, const char *e, double &out)
{
	if (s >= e) return false;
	bool neg = false;
	if (*s == '+' || *s == '-')
	{
		neg = (*s == '-');
		++s;
	}
	else if (!isDigit(*s))
		return false;
	double m = 0.0;
	int nread = 0;
	while (s != e && isDigit(*s))
	{
		m *= 10;
		m += (int)(*s - '0');
		++s;
		++nread;
	}
	if (nread == 0) return false;
	int exp10 = 0;
	bool more = (s != e);
	if (more)
	{
		if (*s == '.')
		{
			static const double frac[] = { 1.0, 0.1, 0.01, 0.001, 0.0001, 0.00001, 0.000001, 0.0000001 };
			++s;
			int k = 1;
			while (s != e && isDigit(*s))
			{
				m += (int)(*s - '0') * (k < 8 ? frac[k] : std::pow(10.0, -k));
				++k;
				++s;
			}
		}
		else if (*s != 'e' && *s != 'E')
			more = false;
	}
	if (more && s != e && (*s == 'e' || *s == 'E'))
	{
		++s;
		bool eneg = false;
		if (s != e && (*s == '+' || *s == '-'))
		{
			eneg = (*s == '-');
			++s;
		}
		else if (s == e || !isDigit(*s))
			return false;
		int n = 0;
		while (s != e && isDigit(*s))
		{
			exp10 = exp10 * 10 + (int)(*s - '0');
			++s;
			++n;
		}
		if (eneg) exp10 = -exp10;
		if (n == 0) return false;
	}
	double v = exp10 ? std::ldexp(m * std::pow(5.0, exp10), exp10) : m;
	out = (neg ? -1 : 1) * v;
	return true;
}

float readReal(Cursor &c, double dflt = 0.0)
{
	const char *b, *e;
	c.token(b, e);
	double v = dflt;
	parseNumber(b, e, v);
	return (float)v;
}

int resolveIndex(int idx, int n)
{
	if (idx > 0) return idx - 1;
	if (idx == 0) return 0;
	return n + idx;
}

struct Corner
{
	int v = -1, vt = -1, vn = -1;
};

void skipIndexField(Cursor &c)
{
	while (c.p < c.end && *c.p != '/' && *c.p != ' ' && *c.p != '\t' && *c.p != '\r') ++c.p;
}

Corner readCorner(Cursor &c, int nv, int nvn, int nvt)
{
	Corner k;
	k.v = resolveIndex(atoi(c.p), nv);
	skipIndexField(c);
	if (c.done() || *c.p != '/') return k;
	++c.p;
	if (!c.done() && *c.p == '/')
	{
		++c.p;
		k.vn = resolveIndex(atoi(c.p), nvn);
		skipIndexField(c);
		return k;
	}
	k.vt = resolveIndex(atoi(c.p), nvt);
	skipIndexField(c);
	if (c.done() || *c.p != '/') return k;
	++c.p;
	k.vn = resolveIndex(atoi(c.p), nvn);
	skipIndexField(c);
	return k;
}

// Splits text into lines on \n, \r\n or bare \r.
std::vector<std::string> splitLines(const std::string &text)
{
	std::vector<std::string> lines;
	std::string cur;
	for (size_t i = 0; i < text.size(); ++i)
	{
		char ch = text[i];
		if (ch == '\n')
		{
			lines.push_back(cur);
			cur.clear();
		}
		else if (ch == '\r')
		{
			if (i + 1 < text.size() && text[i + 1] == '\n') ++i;
			lines.push_back(cur);
			cur.clear();
		}
		else
			cur.push_back(ch);
	}
	if (!cur.empty()) lines.push_back(cur);
	return lines;
}

bool readFile(const std::string &path, std::string &out)
{
	std::ifstream f(path.c_str(), std::ios::binary);
	if (!f) return false;
	std::stringstream ss;
	ss << f.rdbuf();
	out = ss.str();
	return true;
}

std::string firstWord(const char *s)
{
	while (*s == ' ' || *s == '\t' || *s == '\n' || *s == '\v' || *s == '\f' || *s == '\r') ++s;
	const char *b = s;
	while (*s && *s != ' ' && *s != '\t' && *s != '\n' && *s != '\v' && *s != '\f' && *s != '\r') ++s;
	return std::string(b, s);
}

bool isBlank(char c) { return c == ' ' || c == '\t'; }

struct MtlEntry
{
	std::string name;
	float kd[3] = { 0, 0, 0 };
	float ke[3] = { 0, 0, 0 };
	std::string mapKd;  // map_Kd: tinyobj's material_t::diffuse_texname (which the reference never reads)
};

// File name of a `map_Kd [-option args ...] name` statement: options are skipped with the arities of the MTL specification
// (tinyobjloader's ParseTextureNameAndOption), the last plain word is the name.
std::string textureName(const char *t)
{
	std::vector<std::string> words;
	while (*t)
	{
		while (isBlank(*t)) ++t;
		const char *b = t;
		while (*t && !isBlank(*t)) ++t;
		if (t > b) words.push_back(std::string(b, t));
	}
	std::string name;
	for (size_t i = 0; i < words.size(); ++i)
	{
		const std::string &w = words[i];
		if (w == "-blendu" || w == "-blendv" || w == "-clamp" || w == "-imfchan" || w == "-type" || w == "-boost" || w == "-bm" ||
		    w == "-texres")
			i += 1;
		else if (w == "-mm") i += 2;
		else if (w == "-o" || w == "-s" || w == "-t")
		{
			// one to three numbers; the last word of the statement is always the name
			for (int k = 0; k < 3 && i + 2 < words.size() && strchr("+-.0123456789", words[i + 1][0]); ++k) ++i;
		}
		else name = w;
	}
	return name;
}

void parseMtl(const std::string &text, std::vector<MtlEntry> &mats, std::map<std::string, int> &byName)
{
	MtlEntry cur;
	for (std::string line : splitLines(text))
	{
		size_t last = line.find_last_not_of(" \t");
		line = (last == std::string::npos) ? std::string() : line.substr(0, last + 1);
		if (line.empty()) continue;
		const char *t = line.c_str();
		while (isBlank(*t)) ++t;
		if (*t == '\0' || *t == '#') continue;
		if (strncmp(t, "newmtl", 6) == 0 && isBlank(t[6]))
		{
			if (!cur.name.empty())
			{
				byName.insert(std::make_pair(cur.name, (int)mats.size()));
				mats.push_back(cur);
			}
			cur = MtlEntry();
			cur.name = firstWord(t + 7);
			continue;
		}
		if (t[0] == 'K' && (t[1] == 'd' || t[1] == 'e') && isBlank(t[2]))
		{
			Cursor c{ t + 2, t + strlen(t) };
			float *dst = (t[1] == 'd') ? cur.kd : cur.ke;
			dst[0] = readReal(c);
			dst[1] = readReal(c);
			dst[2] = readReal(c);
			continue;
		}
		if (strncmp(t, "map_Kd", 6) == 0 && isBlank(t[6]))
		{
			cur.mapKd = textureName(t + 7);
			continue;
		}
	}
	byName.insert(std::make_pair(cur.name, (int)mats.size()));
	mats.push_back(cur);
}

struct Shape
{
	std::vector<Corner> corners;   // 3 per triangle
	std::vector<int> materialIds;  // 1 per triangle
};

bool flushFaces(Shape &shape, const std::vector<std::vector<Corner>> &faces, int material)
{
	if (faces.empty()) return false;
	for (const auto &face : faces)
	{
		for (size_t k = 2; k < face.size(); ++k)
		{
			shape.corners.push_back(face[0]);
			shape.corners.push_back(face[k - 1]);
			shape.corners.push_back(face[k]);
			shape.materialIds.push_back(material);
		}
	}
	return true;
}

// Vertex identity as the reference's unordered_map<Vertex,...> sees it: float ==, so that
// -0.0f and +0.0f coincide (main.cpp:109-112).  Key = bit pattern with -0 folded to +0.
struct VertexKey
{
	uint32_t w[11];
	bool operator==(const VertexKey &o) const { return memcmp(w, o.w, sizeof(w)) == 0; }
};
struct VertexKeyHash
{
	size_t operator()(const VertexKey &k) const
	{
		uint64_t h = 1469598103934665603ull;
		for (uint32_t x : k.w)
		{
			h ^= x;
			h *= 1099511628211ull;
		}
		return (size_t)h;
	}
};
VertexKey makeKey(const PtVertex &v)
{
	VertexKey k;
	memcpy(k.w, &v, sizeof(k.w));
	for (uint32_t &x : k.w)
		if (x == 0x80000000u) x = 0;
	return k;
}
}  // namespace

namespace
{
// PathTracer::loadModel's loop over what tinyobj::LoadObj returned (main.cpp:1122-1189): materials -> Material{diffuse,
// emission} + the appended grey(0.5) fallback, per-face material indices (-1 -> fallback), first-occurrence vertex
// de-duplication, white vertex colour, V flipped.  corners = tinyobj::index_t per face corner, three per triangle, all
// shapes concatenated in shape order.  Extension: a corner without a normal gets its triangle's flat geometric normal.
int loadModelFromArrays(const char *who, const std::vector<float> &v, const std::vector<float> &vn, const std::vector<float> &vt,
                        const std::vector<Corner> &corners, const std::vector<int> &materialIds,
                        const std::vector<PtMaterial> &mtlMaterials, PtSceneData *sd)
{
	sd->materials = mtlMaterials;
	const uint32_t matCount = (uint32_t)sd->materials.size();
	PtMaterial grey = { { 0.5f, 0.5f, 0.5f }, { 0.0f, 0.0f, 0.0f } };
	sd->materials.push_back(grey);

	std::unordered_map<VertexKey, uint32_t, VertexKeyHash> unique;
	for (int id : materialIds)
	{
		if (id < -1 || (id >= 0 && (uint32_t)id >= matCount))
		{
			g_seamError = std::string(who) + ": material id out of range";
			return PT_EINVAL;
		}
		sd->matIndices.push_back(id == -1 ? matCount : (uint32_t)id);
	}
	for (size_t tri = 0; tri + 2 < corners.size(); tri += 3)
	{
		float flat[3] = { 0, 0, 1 };
		bool haveFlat = false;
		for (int j = 0; j < 3; ++j)
		{
			const Corner &c = corners[tri + j];
			if (c.v < 0 || (size_t)(3 * (size_t)c.v + 2) >= v.size())
			{
				g_seamError = std::string(who) + ": face references a vertex that does not exist";
				return PT_EIO;
			}
		}
		for (int j = 0; j < 3; ++j)
		{
			const Corner &c = corners[tri + j];
			PtVertex vert;
			for (int k = 0; k < 3; ++k) vert.pos[k] = v[3 * (size_t)c.v + k];
			if (c.vn >= 0 && (size_t)(3 * (size_t)c.vn + 2) < vn.size())
			{
				for (int k = 0; k < 3; ++k) vert.normal[k] = vn[3 * (size_t)c.vn + k];
			}
			else
			{
				// Extension (reference: out-of-bounds read, main.cpp:1162-1164): flat normal.
				if (!haveFlat)
				{
					const float *p0 = &v[3 * (size_t)corners[tri].v], *p1 = &v[3 * (size_t)corners[tri + 1].v],
					            *p2 = &v[3 * (size_t)corners[tri + 2].v];
					float e1[3] = { p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2] };
					float e2[3] = { p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2] };
					float cx = e1[1] * e2[2] - e2[1] * e1[2];
					float cy = e1[2] * e2[0] - e2[2] * e1[0];
					float cz = e1[0] * e2[1] - e2[0] * e1[1];
					float l2 = (cx * cx + cy * cy) + cz * cz;
					if (l2 > 0.0f)
					{
						float inv = 1.0f / std::sqrt(l2);
						flat[0] = cx * inv;
						flat[1] = cy * inv;
						flat[2] = cz * inv;
					}
					haveFlat = true;
				}
				for (int k = 0; k < 3; ++k) vert.normal[k] = flat[k];
			}
			if (c.vt >= 0 && (size_t)(2 * (size_t)c.vt + 1) < vt.size())
			{
				vert.texCoord[0] = vt[2 * (size_t)c.vt];
				vert.texCoord[1] = 1.0f - vt[2 * (size_t)c.vt + 1];
			}
			else
			{
				vert.texCoord[0] = 0.0f;
				vert.texCoord[1] = 0.0f;
			}
			vert.color[0] = vert.color[1] = vert.color[2] = 1.0f;
			VertexKey key = makeKey(vert);
			auto it = unique.find(key);
			uint32_t id;
			if (it == unique.end())
			{
				id = (uint32_t)sd->vertices.size();
				unique.emplace(key, id);
				sd->vertices.push_back(vert);
			}
			else
				id = it->second;
			sd->indices.push_back(id);
		}
	}
	return PT_OK;
}
}  // namespace

extern "C" int pt_load_obj(const char *obj_path, PtSceneData **out) try
{
	if (!obj_path || !out)
	{
		g_seamError = "pt_load_obj: null argument";
		return PT_EINVAL;
	}
	std::string text;
	if (!readFile(obj_path, text))
	{
		g_seamError = std::string("pt_load_obj: cannot open ") + obj_path;
		return PT_EIO;
	}
	std::string path(obj_path), mtlDir;
	size_t slash = path.find_last_of("/\\");
	if (slash != std::string::npos) mtlDir = path.substr(0, slash);
	mtlDir += "/";

	std::vector<float> v, vn, vt;
	std::vector<Shape> shapes;
	std::vector<MtlEntry> mtl;
	std::map<std::string, int> mtlByName;
	std::vector<std::vector<Corner>> pending;
	Shape shape;
	int material = -1;
	std::string warnings;

	for (const std::string &line : splitLines(text))
	{
		if (line.empty()) continue;
		const char *t = line.c_str();
		const char *lineEnd = t + line.size();
		while (isBlank(*t)) ++t;
		if (*t == '\0' || *t == '#') continue;
		if (t[0] == 'v' && isBlank(t[1]))
		{
			Cursor c{ t + 2, lineEnd };
			for (int k = 0; k < 3; ++k) v.push_back(readReal(c));
			continue;
		}
		if (t[0] == 'v' && t[1] == 'n' && isBlank(t[2]))
		{
			Cursor c{ t + 3, lineEnd };
			for (int k = 0; k < 3; ++k) vn.push_back(readReal(c));
			continue;
		}
		if (t[0] == 'v' && t[1] == 't' && isBlank(t[2]))
		{
			Cursor c{ t + 3, lineEnd };
			for (int k = 0; k < 2; ++k) vt.push_back(readReal(c));
			continue;
		}
		if (t[0] == 'f' && isBlank(t[1]))
		{
			Cursor c{ t + 2, lineEnd };
			c.skipBlank();
			std::vector<Corner> face;
			while (!c.done() && *c.p != '\r' && *c.p != '\n')
			{
				face.push_back(readCorner(c, (int)(v.size() / 3), (int)(vn.size() / 3), (int)(vt.size() / 2)));
				while (!c.done() && (*c.p == ' ' || *c.p == '\t' || *c.p == '\r')) ++c.p;
			}
			pending.push_back(face);
			continue;
		}
		if (strncmp(t, "usemtl", 6) == 0 && isBlank(t[6]))
		{
			std::string name = firstWord(t + 7);
			auto it = mtlByName.find(name);
			int id = (it == mtlByName.end()) ? -1 : it->second;
			if (id != material)
			{
				flushFaces(shape, pending, material);
				pending.clear();
				material = id;
			}
			continue;
		}
		if (strncmp(t, "mtllib", 6) == 0 && isBlank(t[6]))
		{
			std::string rest(t + 7);
			bool found = false;
			size_t pos = 0;
			while (pos <= rest.size())
			{
				size_t sp = rest.find(' ', pos);
				std::string fn = rest.substr(pos, sp == std::string::npos ? std::string::npos : sp - pos);
				pos = (sp == std::string::npos) ? rest.size() + 1 : sp + 1;
				std::string mtlText;
				if (readFile(mtlDir + fn, mtlText))
				{
					parseMtl(mtlText, mtl, mtlByName);
					found = true;
					break;
				}
				warnings += "WARN: Material file [ " + mtlDir + fn + " ] not found.\n";
			}
			if (!found) warnings += "WARN: Failed to load material file(s). Use default material.\n";
			continue;
		}
		if ((t[0] == 'g' || t[0] == 'o') && isBlank(t[1]))
		{
			if (flushFaces(shape, pending, material)) shapes.push_back(shape);
			shape = Shape();
			pending.clear();
			continue;
		}
	}
	if (flushFaces(shape, pending, material) || !shape.corners.empty()) shapes.push_back(shape);

	// ---- PathTracer::loadModel (main.cpp:1122-1189) ----
	std::vector<PtMaterial> mats;
	for (const MtlEntry &m : mtl)
	{
		PtMaterial pm;
		for (int k = 0; k < 3; ++k)
		{
			pm.diffuse[k] = m.kd[k];
			pm.emission[k] = m.ke[k];
		}
		mats.push_back(pm);
	}
	std::vector<Corner> corners;
	std::vector<int> materialIds;
	for (const Shape &s : shapes)
	{
		corners.insert(corners.end(), s.corners.begin(), s.corners.end());
		materialIds.insert(materialIds.end(), s.materialIds.begin(), s.materialIds.end());
	}
	PtSceneData *sd = new PtSceneData();
	sd->warnings = warnings;
	for (const MtlEntry &m : mtl)
		sd->diffuseMaps.push_back(m.mapKd.empty() || m.mapKd[0] == '/' ? m.mapKd : mtlDir + m.mapKd);
	int rc = loadModelFromArrays("pt_load_obj", v, vn, vt, corners, materialIds, mats, sd);
	if (rc)
	{
		delete sd;
		return rc;
	}
	*out = sd;
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	g_seamError = "pt_load_obj: out of host memory";
	return PT_ENOMEM;
}

extern "C" int pt_scene_from_tinyobj(const float *vertices, uint32_t n_vertices, const float *normals, uint32_t n_normals,
                                    const float *texcoords, uint32_t n_texcoords, const int32_t *indices,
                                    uint32_t n_indices, const int32_t *material_ids, const PtMaterial *materials,
                                    uint32_t n_materials, PtSceneData **out) try
{
	if (!out || (n_vertices && !vertices) || (n_normals && !normals) || (n_texcoords && !texcoords) || (n_indices && !indices) ||
	    (n_indices && !material_ids) || (n_materials && !materials) || n_indices % 3)
	{
		g_seamError = "pt_scene_from_tinyobj: bad argument";
		return PT_EINVAL;
	}
	std::vector<float> v(vertices, vertices + 3 * (size_t)n_vertices), vn(normals, normals + 3 * (size_t)n_normals),
	    vt(texcoords, texcoords + 2 * (size_t)n_texcoords);
	std::vector<Corner> corners(n_indices);
	for (uint32_t i = 0; i < n_indices; ++i)
	{
		// tinyobj::index_t { vertex_index, normal_index, texcoord_index }
		corners[i].v = indices[3 * (size_t)i];
		corners[i].vn = indices[3 * (size_t)i + 1];
		corners[i].vt = indices[3 * (size_t)i + 2];
	}
	std::vector<int> ids(material_ids, material_ids + n_indices / 3);
	std::vector<PtMaterial> mats(materials, materials + n_materials);
	PtSceneData *sd = new PtSceneData();
	int rc = loadModelFromArrays("pt_scene_from_tinyobj", v, vn, vt, corners, ids, mats, sd);
	if (rc)
	{
		delete sd;
		return rc;
	}
	*out = sd;
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	g_seamError = "pt_scene_from_tinyobj: out of host memory";
	return PT_ENOMEM;
}

extern "C" void pt_scene_free(PtSceneData *s) { delete s; }
extern "C" uint32_t pt_scene_n_vertices(const PtSceneData *s) { return (uint32_t)s->vertices.size(); }
extern "C" uint32_t pt_scene_n_indices(const PtSceneData *s) { return (uint32_t)s->indices.size(); }
extern "C" uint32_t pt_scene_n_materials(const PtSceneData *s) { return (uint32_t)s->materials.size(); }
extern "C" const PtVertex *pt_scene_vertices(const PtSceneData *s) { return s->vertices.data(); }
extern "C" const uint32_t *pt_scene_indices(const PtSceneData *s) { return s->indices.data(); }
extern "C" const PtMaterial *pt_scene_materials(const PtSceneData *s) { return s->materials.data(); }
extern "C" const uint32_t *pt_scene_mat_indices(const PtSceneData *s) { return s->matIndices.data(); }
extern "C" const char *pt_scene_warnings(const PtSceneData *s) { return s->warnings.c_str(); }
extern "C" const char *pt_scene_diffuse_map(const PtSceneData *s, uint32_t material)
{
	return (s && material < s->diffuseMaps.size()) ? s->diffuseMaps[material].c_str() : "";
}

// ---------------------------------------------------------------------------- camera ----
// Column-major float[16] (element (column c, row r) at [4c + r]), the memory layout of glm::mat4
// that the reference memcpy's into its uniform buffer (main.cpp:1479).  Arithmetic follows glm
// 1.0.1 operation by operation so the matrices are bit-identical to the reference's:
// lookAtRH (glm/ext/matrix_transform.inl:153-173), perspectiveRH_NO
// (glm/ext/matrix_clip_space.inl:249-262 -- see SURVEY.md section 0 for why NO and not ZO),
// compute_inverse<4,4> (glm/detail/func_matrix.inl:347-404).
namespace
{
struct V3
{
	float x, y, z;
};
inline float dotv(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline V3 crossv(V3 a, V3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
inline V3 unitv(V3 a)
{
	float inv = 1.0f / std::sqrt(dotv(a, a));
	return { a.x * inv, a.y * inv, a.z * inv };
}
}  // namespace

extern "C" void pt_look_at(const float eye[3], const float center[3], const float up[3], float m[16])
{
	V3 e = { eye[0], eye[1], eye[2] };
	V3 fwd = unitv({ center[0] - eye[0], center[1] - eye[1], center[2] - eye[2] });
	V3 side = unitv(crossv(fwd, { up[0], up[1], up[2] }));
	V3 upv = crossv(side, fwd);
	const float cols[16] = { side.x, upv.x, -fwd.x, 0.0f, side.y, upv.y, -fwd.y, 0.0f,
		                     side.z, upv.z, -fwd.z, 0.0f, -dotv(side, e), -dotv(upv, e), dotv(fwd, e), 1.0f };
	memcpy(m, cols, sizeof(cols));
}

extern "C" void pt_perspective(float fovy, float aspect, float zNear, float zFar, float m[16])
{
	const float th = std::tan(fovy / 2.0f);
	memset(m, 0, 16 * sizeof(float));
	m[0] = 1.0f / (aspect * th);
	m[5] = 1.0f / (th);
	m[10] = -(zFar + zNear) / (zFar - zNear);
	m[11] = -1.0f;
	m[14] = -(2.0f * zFar * zNear) / (zFar - zNear);
}

extern "C" void pt_inverse4(const float a[16], float out[16])
{
	auto M = [&](int c, int r) { return a[4 * c + r]; };
	// 2x2 sub-determinants, named s<rows><cols> after the rows/columns of the minor they span
	const float c00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3);
	const float c02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3);
	const float c03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
	const float c04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3);
	const float c06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3);
	const float c07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
	const float c08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2);
	const float c10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2);
	const float c11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
	const float c12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3);
	const float c14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3);
	const float c15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
	const float c16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2);
	const float c18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2);
	const float c19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
	const float c20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1);
	const float c22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1);
	const float c23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
	const float f0[4] = { c00, c00, c02, c03 }, f1[4] = { c04, c04, c06, c07 }, f2[4] = { c08, c08, c10, c11 };
	const float f3[4] = { c12, c12, c14, c15 }, f4[4] = { c16, c16, c18, c19 }, f5[4] = { c20, c20, c22, c23 };
	const float v0[4] = { M(1, 0), M(0, 0), M(0, 0), M(0, 0) }, v1[4] = { M(1, 1), M(0, 1), M(0, 1), M(0, 1) };
	const float v2[4] = { M(1, 2), M(0, 2), M(0, 2), M(0, 2) }, v3[4] = { M(1, 3), M(0, 3), M(0, 3), M(0, 3) };
	float adj[16];
	for (int r = 0; r < 4; ++r)
	{
		const float sa = (r & 1) ? -1.0f : 1.0f, sb = -sa;
		adj[0 + r] = ((v1[r] * f0[r] - v2[r] * f1[r]) + v3[r] * f2[r]) * sa;
		adj[4 + r] = ((v0[r] * f0[r] - v2[r] * f3[r]) + v3[r] * f4[r]) * sb;
		adj[8 + r] = ((v0[r] * f1[r] - v1[r] * f3[r]) + v3[r] * f5[r]) * sa;
		adj[12 + r] = ((v0[r] * f2[r] - v1[r] * f4[r]) + v2[r] * f5[r]) * sb;
	}
	const float det = (M(0, 0) * adj[0] + M(0, 1) * adj[4]) + (M(0, 2) * adj[8] + M(0, 3) * adj[12]);
	const float rdet = 1.0f / det;
	for (int i = 0; i < 16; ++i) out[i] = adj[i] * rdet;
}

extern "C" void pt_reference_camera(uint32_t width, uint32_t height, float view[16], float proj[16])
{
	const float eye[3] = { 0.0f, 1.0f, 3.5f }, center[3] = { 0.0f, 1.0f, 0.0f }, up[3] = { 0.0f, 1.0f, 0.0f };
	pt_look_at(eye, center, up, view);
	const float fovy = 45.0f * 0.01745329251994329576923690768489f;  // glm::radians(45.0f)
	pt_perspective(fovy, width / static_cast<float>(height), 0.1f, 10.0f, proj);
	proj[5] *= -1;  // "Flip y axis" (main.cpp:1477)
}

// ---------------------------------------------------------------------- image writers ----
extern "C" int pt_write_pfm(const char *path, const float *rgba, uint32_t w, uint32_t h) try
{
	FILE *f = fopen(path, "wb");
	if (!f)
	{
		g_seamError = std::string("pt_write_pfm: cannot open ") + path;
		return PT_EIO;
	}
	fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
	std::vector<float> row(3 * (size_t)w);
	for (uint32_t y = 0; y < h; ++y)
	{
		const float *src = rgba + 4 * (size_t)(h - 1 - y) * w;  // PFM stores bottom row first
		for (uint32_t x = 0; x < w; ++x)
			for (int k = 0; k < 3; ++k) row[3 * x + k] = src[4 * x + k];
		fwrite(row.data(), sizeof(float), row.size(), f);
	}
	fclose(f);
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	g_seamError = "pt_write_pfm: out of host memory";
	return PT_ENOMEM;
}

extern "C" int pt_write_ppm(const char *path, const uint8_t *rgba8, uint32_t w, uint32_t h) try
{
	FILE *f = fopen(path, "wb");
	if (!f)
	{
		g_seamError = std::string("pt_write_ppm: cannot open ") + path;
		return PT_EIO;
	}
	fprintf(f, "P6\n%u %u\n255\n", w, h);
	std::vector<uint8_t> row(3 * (size_t)w);
	for (uint32_t y = 0; y < h; ++y)
	{
		for (uint32_t x = 0; x < w; ++x)
			for (int k = 0; k < 3; ++k) row[3 * x + k] = rgba8[4 * ((size_t)y * w + x) + k];
		fwrite(row.data(), 1, row.size(), f);
	}
	fclose(f);
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	g_seamError = "pt_write_ppm: out of host memory";
	return PT_ENOMEM;
}

// ---------------------------------------------------------------------------- PNG ----
// 8-bit RGBA PNG with stored (uncompressed) deflate blocks: no dependency, bit-exact round trip.  The
// reference only presents to a window (main.cpp:1414-1432); SURVEY.md 8(f) row 1 asks for a writer.
namespace
{
struct Crc32Table
{
	uint32_t t[256];
	Crc32Table()
	{
		for (uint32_t i = 0; i < 256; ++i)
		{
			uint32_t c = i;
			for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
			t[i] = c;
		}
	}
};

uint32_t crc32Update(uint32_t crc, const uint8_t *p, size_t n)
{
	static const Crc32Table table;  // initialised once, thread-safe (C++11 function-local static)
	for (size_t i = 0; i < n; ++i) crc = table.t[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
	return crc;
}

void put32be(std::vector<uint8_t> &v, uint32_t x)
{
	v.push_back((uint8_t)(x >> 24));
	v.push_back((uint8_t)(x >> 16));
	v.push_back((uint8_t)(x >> 8));
	v.push_back((uint8_t)x);
}

bool writeChunk(FILE *f, const char type[4], const std::vector<uint8_t> &data)
{
	std::vector<uint8_t> head;
	put32be(head, (uint32_t)data.size());
	head.insert(head.end(), type, type + 4);
	uint32_t crc = crc32Update(0xFFFFFFFFu, head.data() + 4, 4);
	crc = crc32Update(crc, data.data(), data.size()) ^ 0xFFFFFFFFu;
	std::vector<uint8_t> tail;
	put32be(tail, crc);
	return fwrite(head.data(), 1, head.size(), f) == head.size() &&
	       (data.empty() || fwrite(data.data(), 1, data.size(), f) == data.size()) &&
	       fwrite(tail.data(), 1, 4, f) == 4;
}
}  // namespace

extern "C" int pt_write_png(const char *path, const uint8_t *rgba8, uint32_t w, uint32_t h) try
{
	if (!path || !rgba8 || w == 0 || h == 0 || (uint64_t)w * h * 4 + h > 0x7FFFFFF0ull)
	{
		g_seamError = "pt_write_png: bad argument";
		return PT_EINVAL;
	}
	FILE *f = fopen(path, "wb");
	if (!f)
	{
		g_seamError = std::string("pt_write_png: cannot open ") + path;
		return PT_EIO;
	}
	static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
	bool ok = fwrite(sig, 1, 8, f) == 8;
	std::vector<uint8_t> ihdr;
	put32be(ihdr, w);
	put32be(ihdr, h);
	const uint8_t fmt[5] = { 8, 6, 0, 0, 0 };  // 8 bits, RGBA, deflate, adaptive filtering (type 0 used), no interlace
	ihdr.insert(ihdr.end(), fmt, fmt + 5);
	ok = ok && writeChunk(f, "IHDR", ihdr);
	const uint8_t srgb[1] = { 0 };  // rendering intent: the bytes are sRGB-encoded (pt_read_image_srgb8)
	ok = ok && writeChunk(f, "sRGB", std::vector<uint8_t>(srgb, srgb + 1));
	// raw scanlines: filter byte 0 + row
	const size_t stride = (size_t)w * 4 + 1, rawLen = stride * h;
	std::vector<uint8_t> raw(rawLen);
	for (uint32_t y = 0; y < h; ++y)
	{
		raw[y * stride] = 0;
		memcpy(&raw[y * stride + 1], rgba8 + (size_t)y * w * 4, (size_t)w * 4);
	}
	uint32_t s1 = 1, s2 = 0;  // adler32
	for (size_t i = 0; i < rawLen;)
	{
		size_t n = rawLen - i < 5552 ? rawLen - i : 5552;
		for (size_t k = 0; k < n; ++k)
		{
			s1 += raw[i + k];
			s2 += s1;
		}
		s1 %= 65521u;
		s2 %= 65521u;
		i += n;
	}
	std::vector<uint8_t> z;
	z.reserve(rawLen + rawLen / 65535 * 5 + 16);
	z.push_back(0x78);
	z.push_back(0x01);
	for (size_t i = 0; i < rawLen;)
	{
		size_t n = rawLen - i < 65535 ? rawLen - i : 65535;
		z.push_back(i + n == rawLen ? 1 : 0);  // BFINAL, BTYPE = 00 (stored)
		z.push_back((uint8_t)(n & 0xFF));
		z.push_back((uint8_t)(n >> 8));
		z.push_back((uint8_t)(~n & 0xFF));
		z.push_back((uint8_t)((~n >> 8) & 0xFF));
		z.insert(z.end(), raw.begin() + (long)i, raw.begin() + (long)(i + n));
		i += n;
	}
	put32be(z, (s2 << 16) | s1);
	ok = ok && writeChunk(f, "IDAT", z);
	ok = ok && writeChunk(f, "IEND", std::vector<uint8_t>());
	ok = (fclose(f) == 0) && ok;
	if (!ok)
	{
		g_seamError = std::string("pt_write_png: write failed: ") + path;
		return PT_EIO;
	}
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	g_seamError = "pt_write_png: out of host memory";
	return PT_ENOMEM;
}

// ------------------------------------------------------------------------------ PNG reader ----
// What stb_image does for the reference's lineage (textures/viking_room.png).  Own inflate (RFC 1951: stored,
// fixed and dynamic Huffman blocks, canonical codes decoded bit by bit through count/symbol tables) + PNG
// unfiltering (RFC 2083 filters 0-4).  8-bit samples, non-interlaced; colour types 0, 2, 3, 4, 6.
namespace
{
struct BitReader
{
	const uint8_t *p;
	size_t n, pos = 0;
	uint32_t acc = 0;
	int cnt = 0;
	bool bad = false;
	uint32_t bits(int k)
	{
		while (cnt < k)
		{
			if (pos >= n)
			{
				bad = true;
				return 0;
			}
			acc |= (uint32_t)p[pos++] << cnt;
			cnt += 8;
		}
		uint32_t v = acc & ((k < 32 ? (1u << k) : 0u) - 1u);
		acc >>= k;
		cnt -= k;
		return v;
	}
};

struct Huffman
{
	uint16_t count[16];
	uint16_t symbol[288];
	// canonical code from the code lengths; false if the lengths over-subscribe the code space
	bool build(const uint8_t *len, int n)
	{
		memset(count, 0, sizeof(count));
		for (int i = 0; i < n; ++i) ++count[len[i]];
		int left = 1;
		for (int l = 1; l < 16; ++l)
		{
			left = (left << 1) - count[l];
			if (left < 0) return false;
		}
		uint16_t offs[16];
		offs[1] = 0;
		for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
		for (int i = 0; i < n; ++i)
			if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
		return true;
	}
	int decode(BitReader &br) const
	{
		int code = 0, first = 0, index = 0;
		for (int l = 1; l < 16; ++l)
		{
			code |= (int)br.bits(1);
			if (br.bad) return -1;
			int c = count[l];
			if (code - c < first) return symbol[index + (code - first)];
			index += c;
			first = (first + c) << 1;
			code <<= 1;
		}
		return -1;
	}
};

// maxOut: the caller knows the decoded size (PNG: rows x (stride + 1)); anything longer is a corrupt or hostile stream
bool inflateBlocks(const uint8_t *src, size_t n, std::vector<uint8_t> &out, size_t maxOut)
{
	static const uint16_t lenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
	static const uint8_t lenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
	static const uint16_t distBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577 };
	static const uint8_t distExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };
	static const uint8_t clOrder[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
	BitReader br{ src, n };
	for (;;)
	{
		const uint32_t final = br.bits(1), type = br.bits(2);
		if (br.bad || type == 3) return false;
		if (type == 0)
		{
			br.acc = 0;
			br.cnt = 0;  // to the next byte boundary
			if (br.pos + 4 > n) return false;
			const uint32_t len = src[br.pos] | (src[br.pos + 1] << 8), nlen = src[br.pos + 2] | (src[br.pos + 3] << 8);
			br.pos += 4;
			if ((len ^ 0xFFFFu) != nlen || br.pos + len > n || out.size() + len > maxOut) return false;
			out.insert(out.end(), src + br.pos, src + br.pos + len);
			br.pos += len;
		}
		else
		{
			Huffman lit, dist;
			uint8_t lens[320];
			if (type == 1)
			{
				int i = 0;
				for (; i < 144; ++i) lens[i] = 8;
				for (; i < 256; ++i) lens[i] = 9;
				for (; i < 280; ++i) lens[i] = 7;
				for (; i < 288; ++i) lens[i] = 8;
				lit.build(lens, 288);
				for (i = 0; i < 30; ++i) lens[i] = 5;
				dist.build(lens, 30);
			}
			else
			{
				const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
				if (br.bad || nlen > 286 || ndist > 30) return false;
				uint8_t cl[19] = { 0 };
				for (int i = 0; i < ncode; ++i) cl[clOrder[i]] = (uint8_t)br.bits(3);
				Huffman clh;
				if (br.bad || !clh.build(cl, 19)) return false;
				int i = 0;
				while (i < nlen + ndist)
				{
					int sym = clh.decode(br);
					if (sym < 0) return false;
					if (sym < 16) lens[i++] = (uint8_t)sym;
					else
					{
						uint8_t prev = 0;
						int rep;
						if (sym == 16)
						{
							if (i == 0) return false;
							prev = lens[i - 1];
							rep = 3 + (int)br.bits(2);
						}
						else if (sym == 17) rep = 3 + (int)br.bits(3);
						else rep = 11 + (int)br.bits(7);
						if (br.bad || i + rep > nlen + ndist) return false;
						while (rep--) lens[i++] = prev;
					}
				}
				if (lens[256] == 0 || !lit.build(lens, nlen) || !dist.build(lens + nlen, ndist)) return false;
			}
			for (;;)
			{
				int sym = lit.decode(br);
				if (sym < 0) return false;
				if (sym < 256)
				{
					if (out.size() >= maxOut) return false;
					out.push_back((uint8_t)sym);
				}
				else if (sym == 256) break;
				else
				{
					sym -= 257;
					if (sym >= 29) return false;
					const size_t len = lenBase[sym] + br.bits(lenExtra[sym]);
					const int ds = dist.decode(br);
					if (ds < 0 || ds >= 30) return false;
					const size_t d = distBase[ds] + br.bits(distExtra[ds]);
					if (br.bad || d > out.size() || out.size() + len > maxOut) return false;
					const size_t from = out.size() - d;
					for (size_t k = 0; k < len; ++k) out.push_back(out[from + k]);  // may overlap: byte by byte
				}
			}
		}
		if (final) return !br.bad;
	}
}

uint32_t get32be(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
}  // namespace

extern "C" void pt_free(void *p) { free(p); }

extern "C" int pt_load_png(const char *path, uint8_t **outRgba, uint32_t *outW, uint32_t *outH) try
{
	if (!path || !outRgba || !outW || !outH)
	{
		g_seamError = "pt_load_png: bad argument";
		return PT_EINVAL;
	}
	*outRgba = nullptr;
	*outW = *outH = 0;
	FILE *f = fopen(path, "rb");
	if (!f)
	{
		g_seamError = std::string("pt_load_png: cannot open ") + path;
		return PT_EIO;
	}
	std::vector<uint8_t> file;
	uint8_t buf[65536];
	size_t got;
	while ((got = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + got);
	fclose(f);
	auto bad = [&](const char *why) {
		g_seamError = std::string("pt_load_png: ") + path + ": " + why;
		return PT_EIO;
	};
	static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
	if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) return bad("not a PNG file");
	uint32_t w = 0, h = 0;
	int colour = -1;
	std::vector<uint8_t> idat, palette, trns;
	bool end = false;
	for (size_t at = 8; at + 12 <= file.size() && !end;)
	{
		const uint32_t len = get32be(&file[at]);
		const uint8_t *type = &file[at + 4], *data = &file[at + 8];
		if (len > file.size() || at + 12 + (size_t)len > file.size()) return bad("truncated chunk");
		const uint32_t crc = crc32Update(0xFFFFFFFFu, type, 4 + (size_t)len) ^ 0xFFFFFFFFu;
		if (crc != get32be(data + len)) return bad("chunk CRC mismatch");
		if (!memcmp(type, "IHDR", 4))
		{
			if (len != 13) return bad("bad IHDR");
			w = get32be(data);
			h = get32be(data + 4);
			colour = data[9];
			if (data[8] != 8) return bad("only 8-bit samples are supported");
			if (data[10] != 0 || data[11] != 0 || data[12] != 0) return bad("interlaced or non-standard PNG");
			if (colour != 0 && colour != 2 && colour != 3 && colour != 4 && colour != 6) return bad("bad colour type");
			if (w == 0 || h == 0 || w > 16384 || h > 16384) return bad("extent out of range (1..16384)");
		}
		else if (!memcmp(type, "PLTE", 4)) palette.assign(data, data + len);
		else if (!memcmp(type, "tRNS", 4)) trns.assign(data, data + len);
		else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
		else if (!memcmp(type, "IEND", 4)) end = true;
		at += 12 + (size_t)len;
	}
	if (colour < 0 || !end || idat.size() < 6) return bad("missing IHDR, IDAT or IEND");
	if ((idat[0] & 0x0F) != 8 || ((idat[0] << 8) | idat[1]) % 31 != 0 || (idat[1] & 0x20)) return bad("bad zlib header");
	const int ch = colour == 0 ? 1 : colour == 2 ? 3 : colour == 3 ? 1 : colour == 4 ? 2 : 4;
	const size_t stride = (size_t)w * ch, rawLen = (stride + 1) * h;
	std::vector<uint8_t> raw;
	// deflate expands at most 1032 : 1, so a small file that merely CLAIMS a huge extent reserves little
	raw.reserve(std::min<uint64_t>(rawLen, (uint64_t)idat.size() * 1032u + 64u));
	if (!inflateBlocks(idat.data() + 2, idat.size() - 6, raw, rawLen) || raw.size() != rawLen) return bad("corrupt deflate stream");
	uint32_t s1 = 1, s2 = 0;
	for (size_t i = 0; i < rawLen;)
	{
		size_t n = rawLen - i < 5552 ? rawLen - i : 5552;
		for (size_t k = 0; k < n; ++k)
		{
			s1 += raw[i + k];
			s2 += s1;
		}
		s1 %= 65521u;
		s2 %= 65521u;
		i += n;
	}
	if (((s2 << 16) | s1) != get32be(&idat[idat.size() - 4])) return bad("adler32 mismatch");
	// unfilter in place (RFC 2083 section 6): a = left, b = up, c = upper left
	for (uint32_t y = 0; y < h; ++y)
	{
		uint8_t *row = &raw[(stride + 1) * y + 1];
		const uint8_t *up = y ? row - (stride + 1) : nullptr;
		const int filter = row[-1];
		if (filter > 4) return bad("bad filter type");
		for (size_t i = 0; i < stride; ++i)
		{
			const int a = i >= (size_t)ch ? row[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
			int pred = 0;
			if (filter == 1) pred = a;
			else if (filter == 2) pred = b;
			else if (filter == 3) pred = (a + b) >> 1;
			else if (filter == 4)
			{
				const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
				pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
			}
			row[i] = (uint8_t)(row[i] + pred);
		}
	}
	uint8_t *out = (uint8_t *)malloc((size_t)w * h * 4);
	if (!out)
	{
		g_seamError = "pt_load_png: out of memory";
		return PT_ENOMEM;
	}
	for (uint32_t y = 0; y < h; ++y)
	{
		const uint8_t *row = &raw[(stride + 1) * y + 1];
		uint8_t *dst = out + (size_t)y * w * 4;
		for (uint32_t x = 0; x < w; ++x, dst += 4)
		{
			const uint8_t *s = row + (size_t)x * ch;
			if (colour == 0) { dst[0] = dst[1] = dst[2] = s[0]; dst[3] = 255; }
			else if (colour == 2) { dst[0] = s[0]; dst[1] = s[1]; dst[2] = s[2]; dst[3] = 255; }
			else if (colour == 4) { dst[0] = dst[1] = dst[2] = s[0]; dst[3] = s[1]; }
			else if (colour == 6) { dst[0] = s[0]; dst[1] = s[1]; dst[2] = s[2]; dst[3] = s[3]; }
			else
			{
				const size_t k = s[0];
				if (3 * k + 2 >= palette.size())
				{
					free(out);
					return bad("palette index out of range");
				}
				dst[0] = palette[3 * k];
				dst[1] = palette[3 * k + 1];
				dst[2] = palette[3 * k + 2];
				dst[3] = k < trns.size() ? trns[k] : 255;
			}
		}
	}
	*outRgba = out;
	*outW = w;
	*outH = h;
	return PT_OK;
}
catch (const std::bad_alloc &)
{
	g_seamError = "pt_load_png: out of host memory";
	return PT_ENOMEM;
}

namespace
{
struct SrgbTable
{
	float lut[256];
	SrgbTable()
	{
		for (int i = 0; i < 256; ++i)
		{
			const double c = (double)i / 255.0;
			lut[i] = (float)(c <= 0.04045 ? c / 12.92 : pow((c + 0.055) / 1.055, 2.4));
		}
	}
};
}  // namespace

extern "C" void pt_srgb8_to_linear(const uint8_t *srgb8, uint32_t n, float *out)
{
	static const SrgbTable table;  // initialised once, thread-safe (C++11 function-local static)
	for (uint32_t i = 0; i < n; ++i) out[i] = table.lut[srgb8[i]];
}
