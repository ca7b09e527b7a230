/*
 * adapter/kkintegrator_gpu.h -- the Mitsuba-side half of the drop-in.
 *
 * NOT COMPILED IN THIS REPOSITORY: it needs a Mitsuba 0.5 header tree (<mitsuba/render/scene.h> ...), which neither the
 * reference checkout (src/ only) nor the build image holds.  Everything it calls on the library side is built and tested
 * here (include/kkdirect.h; the parameter extraction it relies on is kkd_parse_bsdf_stream, tests/test_adapter_params.py),
 * and tests/test_adapter_params.py checks every Mitsuba-side name it uses against the reference's sources.
 *
 * How it attaches to the class that exists: DirectIntegrator is defined INSIDE src/integrators/kk/direct.cpp (:15-196;
 * there is no direct.h although :2 includes one), with private members brdfSamplesNo / lightSamplesNo / hideEmitters
 * (:192-194).  So the drop-in is an ADDITION to that class, three lines in kk/direct.cpp (INTEGRATION.md shows the diff):
 *     - #include "direct.h"                      (:2; the file does not exist)
 *     + #include "kkintegrator_gpu.h"            (this file: free functions, no class)
 *     + #include "kk_direct_additions.inl"       (inside the class, before MTS_DECLARE_CLASS(), :180: render() / cancel())
 * Li(), the constructors, serialize() (size, size, bool: :31-36) and the plugin exports (:198-199) stay as they are, so the
 * plugin name, the XML parameters, network serialisation and nested use (src/librender/integrator.cpp:55-88) do not change.
 * Build: the 'kkIntegrator' target (src/integrators/SConscript:44) with -I<repo>/include -I<repo>/adapter, link -lkkdirect.
 *
 * What render() then does (src/librender/integrator.cpp:95-129 is the default it overrides; precedent in this fork:
 * src/integrators/kk/vcm.cpp:196-218): flatten the scene through public accessors, render on the GPU(s), hand the
 * ESpectrumAlphaWeight film to Film::put (src/films/hdrfilm.cpp:391-393).
 */
#pragma once
#include <mitsuba/render/scene.h>
#include <mitsuba/render/renderqueue.h>
#include <mitsuba/core/mstream.h>
#include <mitsuba/core/statistics.h>
#include <map>
#include "kkdirect.h"

MTS_NAMESPACE_BEGIN

/* what the class keeps between render() and cancel(): the context of the render in flight (cancel() runs on another thread) */
struct KKDGpuState { kkd_multi *volatile ctx; KKDGpuState() : ctx(NULL) { } };

static void kkdOnProgress(void *user, uint64_t done, uint64_t total) {
    static_cast<ProgressReporter *>(user)->update((size_t) (total ? done * 1000 / total : 1000));
}

/* SamplingIntegrator::cancel for the GPU render (src/librender/integrator.cpp:90-93) */
static inline void kkdCancelOnGpu(KKDGpuState *gpu) { kkd_multi *c = gpu->ctx; if (c) kkd_multi_request_cancel(c); }

/* The body of DirectIntegrator::render(); bsdfSamples / lightSamples / hideEmitters are the class's brdfSamplesNo /
 * lightSamplesNo / hideEmitters (src/integrators/kk/direct.cpp:17-21, 192-194). */
static bool kkdRenderOnGpu(Scene *scene, RenderQueue *queue, const RenderJob *job, size_t bsdfSamples, size_t lightSamples,
                           bool hideEmittersFlag, KKDGpuState *gpu) {
    {
        kkd_multi *&m_ctx = const_cast<kkd_multi *&>(gpu->ctx);
        /* ---- geometry: Scene::getMeshes() holds the TriMesh of every shape in addShape order, which is the order
         * ShapeKDTree numbers primitives in (src/librender/scene.cpp:558-591, src/librender/skdtree.cpp:49-70), so
         * global primitive id = first_prim + primIndex on both sides ---- */
        const std::vector<TriMesh *> &meshes = scene->getMeshes();
        std::vector<float> pos, nrm, uv; std::vector<uint32_t> idx;
        std::vector<kkd_shape> shapes; std::vector<kkd_emitter> emitters;
        bool anyNormals = false, anyUV = false;
        for (size_t s = 0; s < meshes.size(); ++s) { anyNormals |= meshes[s]->hasVertexNormals(); anyUV |= meshes[s]->hasVertexTexcoords(); }

        /* ---- BSDFs: serialise them all with ONE InstanceManager and let the library decode the stream ---- */
        ref<MemoryStream> ms = new MemoryStream();
        ms->setByteOrder(Stream::ELittleEndian);
        ref<InstanceManager> im = new InstanceManager();
        std::map<const BSDF *, int> topLevelOrder;
        for (size_t s = 0; s < meshes.size(); ++s) {
            const BSDF *bsdf = meshes[s]->getBSDF();
            if (!bsdf) SLog(EError, "kkIntegrator: shape %i has no BSDF (index-matched media are not supported)", (int) s);
            if (topLevelOrder.find(bsdf) == topLevelOrder.end()) {
                topLevelOrder[bsdf] = (int) topLevelOrder.size();
                im->serialize(ms, bsdf);
            }
        }
        std::vector<kkd_material> mats(4 * topLevelOrder.size() + 4);
        uint32_t nMats = 0;
        if (kkd_parse_bsdf_stream(ms->getData(), ms->getPos(), &mats[0], (uint32_t) mats.size(), &nMats) != KKD_OK)
            SLog(EError, "kkIntegrator: %s", kkd_last_error(NULL));
        mats.resize(nMats);
        /* the k-th serialised top-level BSDF is the k-th entry that no earlier entry names as 'nested' */
        std::vector<int> topLevelIndex;
        { std::vector<char> nested(nMats, 0);
          for (uint32_t i = 0; i < nMats; ++i) { for (int k = 0; k < 2; ++k) if (mats[i].nested[k] >= 0) nested[mats[i].nested[k]] = 1;
                                                 }
          for (uint32_t i = 0; i < nMats; ++i) if (!nested[i]) topLevelIndex.push_back((int) i); }

        for (size_t s = 0; s < meshes.size(); ++s) {
            const TriMesh *m = meshes[s];
            const uint32_t v0 = (uint32_t) (pos.size() / 3);
            kkd_shape sh; memset(&sh, 0, sizeof(sh));
            sh.first_prim = (uint32_t) (idx.size() / 3); sh.n_prims = (uint32_t) m->getTriangleCount();
            sh.material_id = topLevelIndex[topLevelOrder[m->getBSDF()]]; sh.emitter_id = -1;
            const Point *P = m->getVertexPositions(); const Normal *N = m->getVertexNormals(); const Point2 *T = m->getVertexTexcoords();
            for (size_t v = 0; v < m->getVertexCount(); ++v) {
                pos.push_back(P[v].x); pos.push_back(P[v].y); pos.push_back(P[v].z);
                if (anyNormals) { nrm.push_back(N ? N[v].x : 0); nrm.push_back(N ? N[v].y : 0); nrm.push_back(N ? N[v].z : 0); }
                if (anyUV) { uv.push_back(T ? T[v].x : 0); uv.push_back(T ? T[v].y : 0); }
            }
            if (N) sh.flags |= KKD_SHAPE_HAS_NORMALS;
            if (T) sh.flags |= KKD_SHAPE_HAS_TEXCOORDS;
            const Triangle *tri = m->getTriangles();
            for (size_t t = 0; t < m->getTriangleCount(); ++t) for (int k = 0; k < 3; ++k) idx.push_back(v0 + tri[t].idx[k]);
            shapes.push_back(sh);
        }

        /* ---- emitters in Scene::getEmitters() order = the order of the emitter CDF (src/librender/scene.cpp:376-380) ---- */
        kkd_envmap env; memset(&env, 0, sizeof(env));
        ref<Bitmap> envBitmap;
        const ref_vector<Emitter> &ems = scene->getEmitters();
        for (size_t e = 0; e < ems.size(); ++e) {
            const Emitter *em = ems[e].get();
            const Properties &p = em->getProperties();
            kkd_emitter ke; memset(&ke, 0, sizeof(ke));
            ke.shape_id = -1; ke.sampling_weight = em->getSamplingWeight();
            const std::string plugin = p.getPluginName();
            if (plugin == "area") {
                ke.type = KKD_EMITTER_AREA;
                Float r, g, b; p.getSpectrum("radiance", Spectrum::getD65()).toLinearRGB(r, g, b);
                ke.radiance[0] = r; ke.radiance[1] = g; ke.radiance[2] = b;
                for (size_t s = 0; s < meshes.size(); ++s) if (meshes[s]->getEmitter() == em) { ke.shape_id = (int32_t) s; shapes[s].emitter_id = (int32_t) e; }
                if (ke.shape_id < 0) SLog(EError, "kkIntegrator: area light %i has no triangle-mesh parent", (int) e);
            } else if (plugin == "point") {
                ke.type = KKD_EMITTER_POINT;
                Float r, g, b; p.getSpectrum("intensity", Spectrum::getD65()).toLinearRGB(r, g, b);
                ke.radiance[0] = r; ke.radiance[1] = g; ke.radiance[2] = b;
                const Point o = em->getWorldTransform()->eval(0)(Point(0.0f));
                ke.position[0] = o.x; ke.position[1] = o.y; ke.position[2] = o.z;
            } else if (plugin == "envmap") {
                ke.type = KKD_EMITTER_ENVMAP;
                envBitmap = em->getBitmap()->convert(Bitmap::ERGB, Bitmap::EFloat32);      /* level 0 of the MIP map */
                env.width = envBitmap->getWidth(); env.height = envBitmap->getHeight(); env.texels = envBitmap->getFloat32Data();
                env.scale = p.getFloat("scale", 1.0f);
                const Matrix4x4 &mtx = em->getWorldTransform()->eval(0).getMatrix();
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) env.to_world[3 * r + c] = mtx(r, c);
                BSphere bs(scene->getAABB().getBSphere());                                   /* src/emitters/envmap.cpp:333-334 */
                bs.radius = std::max(Epsilon, bs.radius * 1.5f);
                env.bsphere_center[0] = bs.center.x; env.bsphere_center[1] = bs.center.y; env.bsphere_center[2] = bs.center.z;
                env.bsphere_radius = bs.radius;
            } else
                SLog(EError, "kkIntegrator: emitter plugin \"%s\" is not on the kk direct path (area, point, envmap)", plugin.c_str());
            emitters.push_back(ke);
        }

        /* ---- sensor, film, reconstruction filter (src/sensors/perspective.cpp:126-180, src/librender/film.cpp:89-96) ---- */
        const Sensor *sensor = scene->getSensor();
        if (sensor->getProperties().getPluginName() != "perspective")
            SLog(EError, "kkIntegrator: only the perspective sensor is on the kk direct path");
        const PerspectiveCamera *cam = static_cast<const PerspectiveCamera *>(sensor);
        const Film *film = sensor->getFilm();
        kkd_sensor ks; memset(&ks, 0, sizeof(ks));
        const Matrix4x4 &tw = cam->getWorldTransform()->eval(0).getMatrix();
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) ks.to_world[4 * r + c] = tw(r, c);
        ks.xfov_deg = cam->getXFov(); ks.near_clip = cam->getNearClip(); ks.far_clip = cam->getFarClip();
        ks.film_width = film->getSize().x; ks.film_height = film->getSize().y;
        ks.crop_offset_x = film->getCropOffset().x; ks.crop_offset_y = film->getCropOffset().y;
        ks.crop_width = film->getCropSize().x; ks.crop_height = film->getCropSize().y;
        ks.film_has_alpha = film->hasAlpha() ? 1 : 0;
        ks.film_high_quality_edges = film->hasHighQualityEdges() ? 1 : 0;
        const ReconstructionFilter *rf = film->getReconstructionFilter();
        const std::string rfName = rf->getProperties().getPluginName();
        ks.rfilter_type = rfName == "box" ? KKD_RFILTER_BOX : rfName == "tent" ? KKD_RFILTER_TENT : rfName == "mitchell" ? KKD_RFILTER_MITCHELL
                        : rfName == "catmullrom" ? KKD_RFILTER_CATMULLROM : rfName == "lanczos" ? KKD_RFILTER_LANCZOS : KKD_RFILTER_GAUSSIAN;
        ks.rfilter_radius = rf->getRadius();
        ks.rfilter_param[0] = rf->getProperties().getFloat(rfName == "gaussian" ? "stddev" : rfName == "lanczos" ? "lobes" : "B", rfName == "gaussian" ? 0.5f : rfName == "lanczos" ? 3.0f : 1 / 3.0f);
        ks.rfilter_param[1] = rf->getProperties().getFloat("C", 1 / 3.0f);

        kkd_scene_desc sd; memset(&sd, 0, sizeof(sd));
        sd.abi_version = KKD_ABI_VERSION;
        sd.n_verts = pos.size() / 3; sd.n_tris = idx.size() / 3;
        sd.positions = &pos[0]; sd.indices = &idx[0];
        sd.normals = anyNormals ? &nrm[0] : NULL; sd.texcoords = anyUV ? &uv[0] : NULL;
        sd.n_shapes = (uint32_t) shapes.size(); sd.shapes = &shapes[0];
        sd.n_materials = (uint32_t) mats.size(); sd.materials = &mats[0];
        sd.n_emitters = (uint32_t) emitters.size(); sd.emitters = emitters.empty() ? NULL : &emitters[0];
        sd.envmap = env; sd.sensor = ks;

        /* ---- render ---- */
        /* every GPU of the node, image tiles dealt out among them (what the scheduler's workers are to the CPU path,
         * src/librender/integrator.cpp:102-109); a single GPU is the same call with one device */
        std::vector<int32_t> devices(std::max(1, kkd_device_count()));
        for (size_t i = 0; i < devices.size(); ++i) devices[i] = (int32_t) i;
        kkd_multi_config cfg; memset(&cfg, 0, sizeof(cfg));
        cfg.n_devices = (uint32_t) devices.size(); cfg.devices = &devices[0]; cfg.split = KKD_SPLIT_TILE; cfg.seed = 1;
        m_ctx = kkd_multi_create(&cfg);
        if (!m_ctx) SLog(EError, "kkIntegrator: %s", kkd_multi_last_error(NULL));    /* throws (src/librender/renderjob.cpp:111-115) */
        if (kkd_multi_scene_upload(m_ctx, &sd) != KKD_OK) { std::string msg = kkd_multi_last_error(m_ctx); kkd_multi_destroy(m_ctx); m_ctx = NULL; SLog(EError, "kkIntegrator: %s", msg.c_str()); }
        SLog(EInfo, "kkIntegrator: rendering on %i GPU(s)", (int) kkd_multi_device_count(m_ctx));

        /* progress: ProgressReporter as BlockedRenderProcess does (src/librender/renderproc.cpp:146, 179) */
        ProgressReporter progress("Rendering", 1000, job);
        kkd_multi_set_progress_callback(m_ctx, &kkdOnProgress, &progress);

        kkd_render_params rp; memset(&rp, 0, sizeof(rp));
        rp.spp = (uint32_t) scene->getSampler()->getSampleCount();                   /* src/librender/integrator.cpp:103-104 */
        rp.bsdf_samples = (uint32_t) bsdfSamples; rp.light_samples = (uint32_t) lightSamples; rp.hide_emitters = hideEmittersFlag ? 1 : 0;
        rp.block_size = scene->getBlockSize(); rp.tile_count = 1;
        /* the scene's sampler plugin: Mitsuba's default ldsampler (src/samplers/ldsampler.cpp; its 'dimension' is not readable
         * from outside the plugin: the default, 4), the stratified sampler (src/samplers/stratified.cpp; its getSampleCount()
         * is already the perfect square) or the independent sampler; other samplers fall back to independent */
        if (scene->getSampler()->getClass()->getName() == "LowDiscrepancySampler") rp.sampler = KKD_SAMPLER_LDSAMPLER;
        else if (scene->getSampler()->getClass()->getName() == "StratifiedSampler") rp.sampler = KKD_SAMPLER_STRATIFIED;
        /* halton: deterministic; its 'scramble' is not readable from outside the plugin: the default, -1 (Faure permutations) */
        else if (scene->getSampler()->getClass()->getName() == "HaltonSampler") rp.sampler = KKD_SAMPLER_HALTON;
        else if (scene->getSampler()->getClass()->getName() == "HammersleySampler") rp.sampler = KKD_SAMPLER_HAMMERSLEY;
        /* sobol: deterministic with its default scramble = 0 (samplers/sobol.cpp:93); a scrambled one cannot be told apart from outside */
        else if (scene->getSampler()->getClass()->getName() == "SobolSampler") rp.sampler = KKD_SAMPLER_SOBOL;
        else if (scene->getSampler()->getClass()->getName() != "IndependentSampler")
            SLog(EWarn, "kkIntegrator: sampler \"%s\" is not available on the GPU path, using independent samples", scene->getSampler()->getClass()->getName().c_str());
        ref<ImageBlock> block = new ImageBlock(Bitmap::ESpectrumAlphaWeight, film->getCropSize(), NULL);
        kkd_stats st;
        const int rc = kkd_multi_render(m_ctx, &rp, block->getBitmap()->getFloat32Data(), &st);
        if (rc == KKD_OK) {
            block->setOffset(film->getCropOffset());
            const_cast<Film *>(film)->put(block);                                    /* src/films/hdrfilm.cpp:391-393 */
            queue->signalRefresh(job);                                               /* src/librender/renderqueue.cpp:119-141 */
            SLog(EInfo, "kkIntegrator (GPU): %.1f Msamples/s, %.1f Mrays/s, %.2f s", st.samples / st.seconds * 1e-6,
                (st.rays_closest + st.rays_shadow) / st.seconds * 1e-6, st.seconds);
        } else if (rc != KKD_ERR_CANCELLED) {
            std::string msg = kkd_multi_last_error(m_ctx); kkd_multi_destroy(m_ctx); m_ctx = NULL;
            SLog(EError, "kkIntegrator: %s", msg.c_str());
        }
        kkd_multi_destroy(m_ctx); m_ctx = NULL;
        return rc == KKD_OK;                                                         /* false on cancel (src/librender/integrator.cpp:128) */
    }

}

MTS_NAMESPACE_END
