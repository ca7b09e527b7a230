/*
 * adapter/kk_direct_additions.inl -- the members the drop-in adds to class DirectIntegrator, #included INSIDE the class
 * body of src/integrators/kk/direct.cpp (before MTS_DECLARE_CLASS(), :180).  The class's own members are used under the
 * names they have there: brdfSamplesNo, lightSamplesNo, hideEmitters (src/integrators/kk/direct.cpp:192-194).
 * Li() (:50-72) is untouched and still serves nested use; render() no longer reaches it.
 */
public:
	/* overrides SamplingIntegrator::render (src/librender/integrator.cpp:95-129) */
	bool render(Scene *scene, RenderQueue *queue, const RenderJob *job,
			int sceneResID, int sensorResID, int samplerResID) {
		return kkdRenderOnGpu(scene, queue, job, brdfSamplesNo, lightSamplesNo, hideEmitters, &m_gpu);
	}

	/* overrides SamplingIntegrator::cancel (src/librender/integrator.cpp:90-93) */
	void cancel() {
		kkdCancelOnGpu(&m_gpu);
		SamplingIntegrator::cancel();
	}
private:
	KKDGpuState m_gpu;
public:
