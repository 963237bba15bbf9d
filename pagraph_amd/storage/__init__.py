from .storage import GraphCacheServer, HostFeatureStore, huge_page_tensor
