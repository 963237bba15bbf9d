from .storage import GraphCacheServer, HostFeatureStore
