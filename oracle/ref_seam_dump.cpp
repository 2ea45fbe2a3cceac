// TEST INFRASTRUCTURE (oracle/_ref recipe) -- not part of the product.
//
// Driver that exercises the reference's OWN vendored dependencies where they lie
// under /root/reference/external (tinyobjloader v1.0.6-era header, glm 1.0.1) so
// that the oracle restatements (oracle/obj_oracle.py, oracle/pt_oracle.c) and the
// product's host seam can be pinned against what the reference would really have
// computed.  main.cpp itself cannot be compiled here (needs <vulkan/vulkan.h> and
// GLFW), so only the two header-only libraries are driven; the include order of
// main.cpp:4-9 is reproduced so glm resolves perspective() exactly as the
// reference does (GLM_FORCE_DEPTH_ZERO_TO_ONE defined too late => RH_NO).
//
// Built by oracle/Makefile target `_ref` with -I pointing into /root/reference;
// nothing from the reference is copied into this repository.  Output: raw binary
// dumps under oracle/_ref/ which oracle/make_golden.py packs into tests/golden/.
#define GLM_ENABLE_EXPERIMENTAL
#include <glm/gtx/hash.hpp>

#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>

#define TINYOBJLOADER_IMPLEMENTATION
#include <tiny_obj_loader.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static void dump(const std::string &path, const void *p, size_t bytes)
{
	FILE *f = fopen(path.c_str(), "wb");
	if (!f) { perror(path.c_str()); exit(2); }
	if (bytes) fwrite(p, 1, bytes, f);
	fclose(f);
}

static int dump_model(const std::string &objPath, const std::string &outPrefix)
{
	tinyobj::attrib_t attrib;
	std::vector<tinyobj::shape_t> shapes;
	std::vector<tinyobj::material_t> materials;
	std::string err;
	std::string mtlDir;
	size_t pos = objPath.find_last_of("/\\");
	if (pos != std::string::npos) mtlDir = objPath.substr(0, pos);
	mtlDir += "/";
	if (!tinyobj::LoadObj(&attrib, &shapes, &materials, &err, objPath.c_str(), mtlDir.c_str()))
	{
		fprintf(stderr, "LoadObj failed: %s\n", err.c_str());
		return 1;
	}
	dump(outPrefix + ".v.f32", attrib.vertices.data(), attrib.vertices.size() * 4);
	dump(outPrefix + ".vn.f32", attrib.normals.data(), attrib.normals.size() * 4);
	dump(outPrefix + ".vt.f32", attrib.texcoords.data(), attrib.texcoords.size() * 4);
	// per shape: index triples (v, vn, vt) and material ids, concatenated with a shape table
	std::vector<int32_t> idx, mids, shapeTab;
	for (const auto &s : shapes)
	{
		shapeTab.push_back((int32_t)s.mesh.indices.size());
		shapeTab.push_back((int32_t)s.mesh.material_ids.size());
		for (const auto &i : s.mesh.indices)
		{
			idx.push_back(i.vertex_index);
			idx.push_back(i.normal_index);
			idx.push_back(i.texcoord_index);
		}
		for (int m : s.mesh.material_ids) mids.push_back(m);
	}
	dump(outPrefix + ".idx.i32", idx.data(), idx.size() * 4);
	dump(outPrefix + ".mid.i32", mids.data(), mids.size() * 4);
	dump(outPrefix + ".shapes.i32", shapeTab.data(), shapeTab.size() * 4);
	std::vector<float> mats;
	for (const auto &m : materials)
	{
		for (int k = 0; k < 3; ++k) mats.push_back(m.diffuse[k]);
		for (int k = 0; k < 3; ++k) mats.push_back(m.emission[k]);
	}
	dump(outPrefix + ".mat.f32", mats.data(), mats.size() * 4);
	fprintf(stderr, "%s: v=%zu vn=%zu vt=%zu shapes=%zu idx=%zu mats=%zu warn=[%s]\n", objPath.c_str(),
	        attrib.vertices.size() / 3, attrib.normals.size() / 3, attrib.texcoords.size() / 2, shapes.size(),
	        idx.size() / 3, materials.size(), err.c_str());
	return 0;
}

// Camera of main.cpp:1465-1480 through the real glm, plus inverse() which the
// raygen shader applies (raytrace.rgen:44-46) -- glm's cofactor algorithm is the
// parity oracle for the hoisted host-side inverse.
static void dump_cameras(const std::string &outPrefix)
{
	const uint32_t dims[][2] = { { 512, 512 }, { 800, 600 }, { 1920, 1080 }, { 3840, 2160 }, { 64, 48 }, { 7, 5 } };
	std::vector<float> out;  // per dim: W, H, view16, proj16, invView16, invProj16
	for (auto &d : dims)
	{
		glm::mat4 view = glm::lookAt(glm::vec3(0.0f, 1.0f, 3.5f), glm::vec3(0.0f, 1.0f, 0.0f), glm::vec3(0.0f, 1.0f, 0.0f));
		glm::mat4 proj = glm::perspective(glm::radians(45.0f), d[0] / static_cast<float>(d[1]), 0.1f, 10.0f);
		proj[1][1] *= -1;
		glm::mat4 iv = glm::inverse(view), ip = glm::inverse(proj);
		out.push_back((float)d[0]);
		out.push_back((float)d[1]);
		const glm::mat4 *ms[4] = { &view, &proj, &iv, &ip };
		for (auto m : ms)
			for (int c = 0; c < 4; ++c)
				for (int r = 0; r < 4; ++r) out.push_back((*m)[c][r]);
	}
	dump(outPrefix + ".cameras.f32", out.data(), out.size() * 4);

	// generic lookAt/perspective/inverse known-answer cases (non-trivial poses)
	std::vector<float> kat;  // eye3 center3 up3 fovy aspect near far | view16 proj16 invView16 invProj16
	const float poses[][13] = {
		{ 0.3f, 1.7f, 4.1f, -0.2f, 0.9f, 0.1f, 0.0f, 1.0f, 0.0f, 0.6f, 1.5f, 0.05f, 50.0f },
		{ -260.0f, 140.0f, 380.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, 0.0f, 0.9f, 1.7777778f, 1.0f, 5000.0f },
		{ 2.0f, -3.0f, 1.0f, 0.5f, 0.25f, -0.75f, 0.1f, 0.2f, 0.97f, 1.2f, 0.75f, 0.01f, 100.0f },
	};
	for (auto &p : poses)
	{
		glm::mat4 view = glm::lookAt(glm::vec3(p[0], p[1], p[2]), glm::vec3(p[3], p[4], p[5]), glm::vec3(p[6], p[7], p[8]));
		glm::mat4 proj = glm::perspective(p[9], p[10], p[11], p[12]);
		proj[1][1] *= -1;
		glm::mat4 iv = glm::inverse(view), ip = glm::inverse(proj);
		for (int k = 0; k < 13; ++k) kat.push_back(p[k]);
		const glm::mat4 *ms[4] = { &view, &proj, &iv, &ip };
		for (auto m : ms)
			for (int c = 0; c < 4; ++c)
				for (int r = 0; r < 4; ++r) kat.push_back((*m)[c][r]);
	}
	dump(outPrefix + ".camkat.f32", kat.data(), kat.size() * 4);
}

int main(int argc, char **argv)
{
	if (argc < 3)
	{
		fprintf(stderr, "usage: %s <reference_root> <out_dir>\n", argv[0]);
		return 2;
	}
	std::string root = argv[1], out = argv[2];
	const char *models[] = { "cornell", "plane", "teapot", "viking_room" };
	int rc = 0;
	for (auto m : models) rc |= dump_model(root + "/models/" + m + ".obj", out + "/" + m);
	dump_cameras(out + "/glm");
	return rc;
}
